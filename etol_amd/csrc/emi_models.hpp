// emi_models.hpp -- per-node model functions for the gfx950 node kernels.
//
// Each model gives, for ONE collocation node, what ePSOPT::dae and
// ePSOPT::integrand_cost (reference src/ePSOPT/ePSOPT.cpp:186-276) obtain by
// calling the user's f_t closures, plus the first and second derivatives that
// the reference gets from ADOL-C (derivatives="automatic", ePSOPT.cpp:64-65).
// Everything is written out analytically; structural zeros are literal so the
// compiler folds them.
//
//   z = (x_0..x_{NS-1}, u_0..u_{NC-1}),  NV = NS+NC
//   f(z)   state derivatives            J[i][v] = df_i/dz_v
//   L(z)   integrand cost               g[v]    = dL/dz_v
//   hess() adds  cL*L_zz + sum_i cf[i]*f_i,zz  into the packed lower triangle
//          H[v*(v+1)/2 + q], q<=v
//   NPATH  path rows the model computes itself (0 for the hand-written models, whose keep-outs come from the
//          record table; generated models carry the user's traced constraint callbacks):
//          path(P, z, t, c, cx, cy)      values and partials w.r.t. the two path states
//          path_hess(P, z, t, mu, h)     h[0..2] += sum_j mu_j (c_j,xx  c_j,xy  c_j,yy)
#pragma once
#ifndef __HIPCC_RTC__   // hiprtc (models compiled at run time, emi_rtc.hip) predeclares the device runtime
#include <hip/hip_runtime.h>
#endif

namespace emi {

#define EMI_MAX_PARAMS 16
#define EMI_DEV __device__ __forceinline__

template <typename T> struct ModelParams { T p[EMI_MAX_PARAMS]; };

EMI_DEV void emi_sincos(double a, double* s, double* c) { sincos(a, s, c); }
EMI_DEV void emi_sincos(float a, float* s, float* c) { sincosf(a, s, c); }

// elementary functions used by generated (traced) models, emi_trace.cpp
EMI_DEV double emi_sin(double a) { return sin(a); }
EMI_DEV double emi_cos(double a) { return cos(a); }
EMI_DEV double emi_tan(double a) { return tan(a); }
EMI_DEV double emi_exp(double a) { return exp(a); }
EMI_DEV double emi_log(double a) { return log(a); }
EMI_DEV double emi_sqrt(double a) { return sqrt(a); }
EMI_DEV double emi_pow(double a, double c) { return pow(a, c); }
EMI_DEV float emi_sin(float a) { return sinf(a); }
EMI_DEV float emi_cos(float a) { return cosf(a); }
EMI_DEV float emi_tan(float a) { return tanf(a); }
EMI_DEV float emi_exp(float a) { return expf(a); }
EMI_DEV float emi_log(float a) { return logf(a); }
EMI_DEV float emi_sqrt(float a) { return sqrtf(a); }
EMI_DEV float emi_pow(float a, float c) { return powf(a, c); }

// ---------------------------------------------------------------------------
// 2-state single integrator: reference etol_psopt_example1.cpp
//   dxdt :116-126 (xdot = u0), dydt :128-138 (ydot = u1),
//   objFunction :101-114 (L = u0^2 + u1^2).   No parameters.
// ---------------------------------------------------------------------------
template <typename T> struct PointMass2D {
    static constexpr int NS = 2, NC = 2, NV = 4, NPARAM = 0, NPATH = 0;
    EMI_DEV static void f(const ModelParams<T>&, const T* z, T, T* fo) {
        fo[0] = z[2];
        fo[1] = z[3];
    }
    EMI_DEV static void jac(const ModelParams<T>&, const T*, T, T (*J)[NV]) {
        for (int i = 0; i < NS; ++i)
            for (int v = 0; v < NV; ++v) J[i][v] = T(0);
        J[0][2] = T(1);
        J[1][3] = T(1);
    }
    EMI_DEV static T cost(const ModelParams<T>&, const T* z, T) {
        return z[2] * z[2] + z[3] * z[3];
    }
    EMI_DEV static void grad(const ModelParams<T>&, const T* z, T, T* g) {
        g[0] = T(0); g[1] = T(0);
        g[2] = T(2) * z[2];
        g[3] = T(2) * z[3];
    }
    EMI_DEV static void hess(const ModelParams<T>&, const T*, T, T cL, const T*, T* H) {
        H[2 * 3 / 2 + 2] += T(2) * cL;   // (u0,u0)
        H[3 * 4 / 2 + 3] += T(2) * cL;   // (u1,u1)
    }
};

// ---------------------------------------------------------------------------
// 6-state planar quadrotor (build-defined).
//   x = (px, pz, theta, vx, vz, omega),  u = (thrust, torque)
//   p = {mass, inertia, gravity, w_thrust, w_torque}
//   pxdot = vx, pzdot = vz, thetadot = omega,
//   vxdot = -(T/m) sin(theta), vzdot = (T/m) cos(theta) - g, omegadot = tau/I
//   L = w_thrust*T^2 + w_torque*tau^2   (control effort, as the reference's
//       example objective, etol_psopt_example1.cpp:101-114, with weights)
// ---------------------------------------------------------------------------
template <typename T> struct Quadrotor2D {
    static constexpr int NS = 6, NC = 2, NV = 8, NPARAM = 5, NPATH = 0;
    EMI_DEV static void f(const ModelParams<T>& P, const T* z, T, T* fo) {
        T s, c;
        emi_sincos(z[2], &s, &c);
        const T a = z[6] / P.p[0];
        fo[0] = z[3];
        fo[1] = z[4];
        fo[2] = z[5];
        fo[3] = -a * s;
        fo[4] = a * c - P.p[2];
        fo[5] = z[7] / P.p[1];
    }
    EMI_DEV static void jac(const ModelParams<T>& P, const T* z, T, T (*J)[NV]) {
        for (int i = 0; i < NS; ++i)
            for (int v = 0; v < NV; ++v) J[i][v] = T(0);
        T s, c;
        emi_sincos(z[2], &s, &c);
        const T im = T(1) / P.p[0];
        const T a = z[6] * im;
        J[0][3] = T(1);
        J[1][4] = T(1);
        J[2][5] = T(1);
        J[3][2] = -a * c;
        J[3][6] = -s * im;
        J[4][2] = -a * s;
        J[4][6] = c * im;
        J[5][7] = T(1) / P.p[1];
    }
    EMI_DEV static T cost(const ModelParams<T>& P, const T* z, T) {
        return P.p[3] * z[6] * z[6] + P.p[4] * z[7] * z[7];
    }
    EMI_DEV static void grad(const ModelParams<T>& P, const T* z, T, T* g) {
        for (int v = 0; v < 6; ++v) g[v] = T(0);
        g[6] = T(2) * P.p[3] * z[6];
        g[7] = T(2) * P.p[4] * z[7];
    }
    EMI_DEV static void hess(const ModelParams<T>& P, const T* z, T, T cL, const T* cf, T* H) {
        T s, c;
        emi_sincos(z[2], &s, &c);
        const T im = T(1) / P.p[0];
        const T a = z[6] * im;
        // f3 = -a s : d2/dth2 = a s, d2/dth dT = -c/m ; f4 = a c - g : d2/dth2 = -a c, d2/dth dT = -s/m
        H[2 * 3 / 2 + 2] += cf[3] * (a * s) + cf[4] * (-a * c);     // (theta,theta)
        H[6 * 7 / 2 + 2] += cf[3] * (-c * im) + cf[4] * (-s * im);  // (T,theta)
        H[6 * 7 / 2 + 6] += T(2) * P.p[3] * cL;                     // (T,T)
        H[7 * 8 / 2 + 7] += T(2) * P.p[4] * cL;                     // (tau,tau)
    }
};

// ---------------------------------------------------------------------------
// 12-state rigid-body fixed wing (build-defined).
//   x = (pn, pe, pd, phi, theta, psi, ub, vb, wb, pr, qr, rr)
//   u = (thrust, aileron, elevator, rudder)
//   p = {mass, Ixx, Iyy, Izz, g, qS (dyn.pressure*area at trim speed), CL0,
//        CLa, CD0, CDk, Cl_da, Cm_de, Cn_dr, Vtrim, damp, w_ctrl}
//   Kinematics: standard 3-2-1 Euler; aerodynamics: lift/drag linear-quadratic
//   in alpha = wb/Vtrim, side force -damp*vb, moments from surfaces with rate
//   damping; diagonal inertia with gyroscopic coupling.
//   L = w_ctrl * (thrust^2 + da^2 + de^2 + dr^2)
// ---------------------------------------------------------------------------
template <typename T> struct FixedWing12 {
    static constexpr int NS = 12, NC = 4, NV = 16, NPARAM = 16, NPATH = 0;
    struct Pre {
        T sph, cph, sth, cth, sps, cps, tth, icth;
    };
    EMI_DEV static Pre pre(const T* z) {
        Pre q;
        emi_sincos(z[3], &q.sph, &q.cph);
        emi_sincos(z[4], &q.sth, &q.cth);
        emi_sincos(z[5], &q.sps, &q.cps);
        q.icth = T(1) / q.cth;
        q.tth = q.sth * q.icth;
        return q;
    }
    EMI_DEV static void f(const ModelParams<T>& P, const T* z, T, T* fo) {
        const Pre q = pre(z);
        const T u = z[6], v = z[7], w = z[8], p = z[9], qq = z[10], r = z[11];
        const T m = P.p[0], Ixx = P.p[1], Iyy = P.p[2], Izz = P.p[3], g = P.p[4];
        const T qS = P.p[5], iV = T(1) / P.p[13], damp = P.p[14];
        // position rates: R_bn * (u,v,w)
        fo[0] = q.cth * q.cps * u + (q.sph * q.sth * q.cps - q.cph * q.sps) * v +
                (q.cph * q.sth * q.cps + q.sph * q.sps) * w;
        fo[1] = q.cth * q.sps * u + (q.sph * q.sth * q.sps + q.cph * q.cps) * v +
                (q.cph * q.sth * q.sps - q.sph * q.cps) * w;
        fo[2] = -q.sth * u + q.sph * q.cth * v + q.cph * q.cth * w;
        // Euler rates
        fo[3] = p + q.tth * (q.sph * qq + q.cph * r);
        fo[4] = q.cph * qq - q.sph * r;
        fo[5] = (q.sph * qq + q.cph * r) * q.icth;
        // forces
        const T al = w * iV;
        const T CL = P.p[6] + P.p[7] * al;
        const T CD = P.p[8] + P.p[9] * CL * CL;
        const T X = z[12] - qS * CD;
        const T Y = -damp * v;
        const T Z = -qS * CL;
        fo[6] = r * v - qq * w - g * q.sth + X / m;
        fo[7] = p * w - r * u + g * q.sph * q.cth + Y / m;
        fo[8] = qq * u - p * v + g * q.cph * q.cth + Z / m;
        // moments
        const T Lm = qS * P.p[10] * z[13] - damp * p;
        const T Mm = qS * P.p[11] * z[14] - damp * qq;
        const T Nm = qS * P.p[12] * z[15] - damp * r;
        fo[9] = ((Iyy - Izz) * qq * r + Lm) / Ixx;
        fo[10] = ((Izz - Ixx) * p * r + Mm) / Iyy;
        fo[11] = ((Ixx - Iyy) * p * qq + Nm) / Izz;
    }
    EMI_DEV static void jac(const ModelParams<T>& P, const T* z, T, T (*J)[NV]) {
        for (int i = 0; i < NS; ++i)
            for (int v = 0; v < NV; ++v) J[i][v] = T(0);
        const Pre q = pre(z);
        const T u = z[6], v = z[7], w = z[8], p = z[9], qq = z[10], r = z[11];
        const T m = P.p[0], Ixx = P.p[1], Iyy = P.p[2], Izz = P.p[3], g = P.p[4];
        const T qS = P.p[5], iV = T(1) / P.p[13], damp = P.p[14];
        const T r01 = q.sph * q.sth * q.cps - q.cph * q.sps;
        const T r02 = q.cph * q.sth * q.cps + q.sph * q.sps;
        const T r11 = q.sph * q.sth * q.sps + q.cph * q.cps;
        const T r12 = q.cph * q.sth * q.sps - q.sph * q.cps;
        // f0
        J[0][3] = r02 * v - r01 * w;
        J[0][4] = -q.sth * q.cps * u + q.sph * q.cth * q.cps * v + q.cph * q.cth * q.cps * w;
        J[0][5] = -q.cth * q.sps * u - r11 * v - r12 * w;
        J[0][6] = q.cth * q.cps; J[0][7] = r01; J[0][8] = r02;
        // f1
        J[1][3] = r12 * v - r11 * w;
        J[1][4] = -q.sth * q.sps * u + q.sph * q.cth * q.sps * v + q.cph * q.cth * q.sps * w;
        J[1][5] = q.cth * q.cps * u + r01 * v + r02 * w;
        J[1][6] = q.cth * q.sps; J[1][7] = r11; J[1][8] = r12;
        // f2
        J[2][3] = q.cph * q.cth * v - q.sph * q.cth * w;
        J[2][4] = -q.cth * u - q.sph * q.sth * v - q.cph * q.sth * w;
        J[2][6] = -q.sth; J[2][7] = q.sph * q.cth; J[2][8] = q.cph * q.cth;
        // f3 = p + tth*(sph q + cph r)
        const T sq = q.sph * qq + q.cph * r;
        const T cq = q.cph * qq - q.sph * r;
        J[3][3] = q.tth * cq;
        J[3][4] = sq * q.icth * q.icth;
        J[3][9] = T(1); J[3][10] = q.tth * q.sph; J[3][11] = q.tth * q.cph;
        // f4 = cph q - sph r
        J[4][3] = -sq;
        J[4][10] = q.cph; J[4][11] = -q.sph;
        // f5 = sq / cth
        J[5][3] = cq * q.icth;
        J[5][4] = sq * q.tth * q.icth;
        J[5][10] = q.sph * q.icth; J[5][11] = q.cph * q.icth;
        // forces
        const T al = w * iV;
        const T CL = P.p[6] + P.p[7] * al;
        const T dCL = P.p[7] * iV;
        const T dCD = T(2) * P.p[9] * CL * dCL;
        // f6 = r v - q w - g sth + (thr - qS CD)/m
        J[6][4] = -g * q.cth;
        J[6][7] = r; J[6][8] = -qq - qS * dCD / m; J[6][10] = -w; J[6][11] = v;
        J[6][12] = T(1) / m;
        // f7 = p w - r u + g sph cth - damp v / m
        J[7][3] = g * q.cph * q.cth; J[7][4] = -g * q.sph * q.sth;
        J[7][6] = -r; J[7][7] = -damp / m; J[7][8] = p; J[7][9] = w; J[7][11] = -u;
        // f8 = q u - p v + g cph cth - qS CL / m
        J[8][3] = -g * q.sph * q.cth; J[8][4] = -g * q.cph * q.sth;
        J[8][6] = qq; J[8][7] = -p; J[8][8] = -qS * dCL / m; J[8][9] = -v; J[8][10] = u;
        // moments
        J[9][9] = -damp / Ixx; J[9][10] = (Iyy - Izz) * r / Ixx; J[9][11] = (Iyy - Izz) * qq / Ixx;
        J[9][13] = qS * P.p[10] / Ixx;
        J[10][9] = (Izz - Ixx) * r / Iyy; J[10][10] = -damp / Iyy; J[10][11] = (Izz - Ixx) * p / Iyy;
        J[10][14] = qS * P.p[11] / Iyy;
        J[11][9] = (Ixx - Iyy) * qq / Izz; J[11][10] = (Ixx - Iyy) * p / Izz; J[11][11] = -damp / Izz;
        J[11][15] = qS * P.p[12] / Izz;
    }
    EMI_DEV static T cost(const ModelParams<T>& P, const T* z, T) {
        return P.p[15] * (z[12] * z[12] + z[13] * z[13] + z[14] * z[14] + z[15] * z[15]);
    }
    EMI_DEV static void grad(const ModelParams<T>& P, const T* z, T, T* g) {
        for (int v = 0; v < 12; ++v) g[v] = T(0);
        for (int v = 12; v < 16; ++v) g[v] = T(2) * P.p[15] * z[v];
    }
    // Second derivatives of the fixed-wing model are not hand-written yet; the
    // host refuses emi_hess_* for this model (EMI_ERR_UNSUPPORTED).
    EMI_DEV static void hess(const ModelParams<T>&, const T*, T, T, const T*, T*) {}
};

}  // namespace emi
