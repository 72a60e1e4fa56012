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
    // Second derivatives: GENERATED, not hand-written -- the equations of f() above written once more on the
    // expression trace of etol_amd/host/emi_trace.cpp with the parameter block as inputs, differentiated twice
    // (tests/harness/etol_harness.cpp harness_fixedwing_hess_body; tests/test_trace.py checks that this text is
    // what the generator emits and compares it with the sympy Hessian of tests/golden/models.json).
    EMI_DEV static void hess(const ModelParams<T>& P, const T* z, T, T cL, const T* cf, T* H) {
        T cc[NS + 1];
        cc[0] = cL;
        for (int i = 0; i < NS; ++i) cc[1 + i] = cf[i];
        // ---- generated body begin ----
        const T v16 = emi_sin(z[3]);
        const T v17 = emi_cos(z[3]);
        const T v18 = emi_sin(z[4]);
        const T v19 = emi_cos(z[4]);
        const T v20 = emi_sin(z[5]);
        const T v21 = emi_cos(z[5]);
        const T v23 = T(1.00000000000000000e+00) / v19;
        const T v24 = v18 * v23;
        const T v32 = T(1.00000000000000000e+00) / P.p[13];
        const T v35 = v17 * v18;
        const T v40 = v16 * v18;
        const T v64 = -v18;
        const T v68 = z[11] * v17;
        const T v69 = z[10] * v16;
        const T v70 = v68 + v69;
        const T v101 = v16 * P.p[4];
        const T v109 = v17 * P.p[4];
        const T v131 = P.p[2] - P.p[3];
        const T v136 = P.p[3] - P.p[1];
        const T v141 = P.p[1] - P.p[2];
        const T v193 = P.p[15] * cc[0];
        const T v203 = cc[12] / P.p[3];
        const T v211 = cc[11] / P.p[2];
        const T v223 = cc[10] / P.p[1];
        const T v267 = -cc[9];
        const T v274 = v109 * cc[9];
        const T v275 = v19 * cc[9];
        const T v276 = P.p[4] * v275;
        const T v282 = -cc[8];
        const T v290 = v101 * cc[8];
        const T v291 = v274 + v290;
        const T v292 = v19 * cc[8];
        const T v293 = P.p[4] * v292;
        const T v301 = -cc[7];
        const T v310 = P.p[4] * v301;
        const T v313 = cc[7] / P.p[0];
        const T v328 = -v313;
        const T v331 = P.p[5] * v328;
        const T v343 = v70 * cc[6];
        const T v344 = v23 * cc[6];
        const T v345 = -cc[5];
        const T v348 = z[10] * cc[5];
        const T v349 = v276 + v348;
        const T v352 = z[11] * v345;
        const T v353 = v293 + v352;
        const T v355 = v70 * cc[4];
        const T v356 = v24 * cc[4];
        const T v357 = v344 + v356;
        const T v358 = v16 * v357;
        const T v360 = z[10] * v357;
        const T v361 = v353 + v360;
        const T v362 = v17 * v357;
        const T v364 = z[11] * v357;
        const T v365 = v349 + v364;
        const T v368 = z[6] * cc[3];
        const T v369 = -v368;
        const T v370 = v310 + v369;
        const T v373 = z[7] * cc[3];
        const T v374 = v19 * v373;
        const T v375 = v361 + v374;
        const T v376 = v16 * v373;
        const T v377 = v291 + v376;
        const T v380 = z[8] * cc[3];
        const T v381 = v19 * v380;
        const T v382 = v365 + v381;
        const T v383 = v17 * v380;
        const T v384 = v377 + v383;
        const T v387 = z[6] * cc[2];
        const T v388 = v20 * v387;
        const T v389 = v384 + v388;
        const T v390 = v19 * v387;
        const T v393 = z[7] * cc[2];
        const T v394 = v40 * v393;
        const T v395 = v390 + v394;
        const T v396 = v20 * v393;
        const T v397 = v21 * v393;
        const T v398 = v382 + v397;
        const T v399 = v17 * v393;
        const T v402 = z[8] * cc[2];
        const T v403 = -v402;
        const T v404 = v35 * v402;
        const T v405 = v395 + v404;
        const T v406 = v20 * v402;
        const T v407 = v21 * v403;
        const T v408 = v375 + v407;
        const T v409 = v16 * v403;
        const T v410 = v399 + v409;
        const T v413 = z[6] * cc[1];
        const T v414 = v21 * v413;
        const T v415 = v389 + v414;
        const T v416 = v19 * v413;
        const T v417 = v410 + v416;
        const T v420 = z[7] * cc[1];
        const T v421 = -v420;
        const T v422 = v40 * v420;
        const T v423 = v417 + v422;
        const T v424 = v21 * v420;
        const T v425 = v396 + v424;
        const T v426 = v18 * v425;
        const T v427 = v408 + v426;
        const T v428 = v16 * v425;
        const T v429 = v370 + v428;
        const T v430 = v20 * v421;
        const T v431 = v398 + v430;
        const T v432 = v17 * v421;
        const T v433 = v405 + v432;
        const T v436 = z[8] * cc[1];
        const T v437 = v35 * v436;
        const T v438 = v423 + v437;
        const T v439 = v21 * v436;
        const T v440 = v406 + v439;
        const T v441 = v18 * v440;
        const T v442 = v431 + v441;
        const T v443 = v17 * v440;
        const T v444 = v429 + v443;
        const T v445 = v20 * v436;
        const T v446 = v427 + v445;
        const T v447 = v16 * v436;
        const T v448 = v433 + v447;
        const T v453 = v23 * v355;
        const T v454 = v444 + v453;
        const T v455 = v18 * v355;
        const T v456 = v343 + v455;
        const T v459 = v23 * v456;
        const T v460 = v459 / v19;
        const T v461 = -v460;
        const T v462 = v415 + v461;
        const T v476 = -v442;
        const T v533 = v17 * v345;
        const T v568 = v16 * v446;
        const T v569 = -v568;
        const T v570 = v17 * v476;
        const T v571 = v569 + v570;
        const T v572 = -v462;
        const T v573 = v18 / v19;
        const T v574 = v18 * v460;
        const T v575 = v574 / v19;
        const T v576 = -v575;
        const T v577 = v454 + v576;
        const T v578 = v456 * v573;
        const T v579 = v23 * v573;
        const T v580 = v355 * v579;
        const T v581 = v572 + v580;
        const T v582 = v18 * v579;
        const T v583 = v19 * v355;
        const T v584 = v578 + v583;
        const T v585 = v19 * v23;
        const T v586 = v582 + v585;
        const T v587 = v19 * v440;
        const T v592 = v19 * v425;
        const T v624 = v64 * v380;
        const T v625 = v587 + v624;
        const T v630 = v64 * v373;
        const T v631 = v592 + v630;
        const T v642 = cc[4] * v586;
        const T v644 = cc[6] * v579;
        const T v645 = v642 + v644;
        const T v650 = v64 * cc[8];
        const T v652 = v64 * cc[9];
        const T v654 = P.p[4] * v652;
        const T v655 = v625 + v654;
        const T v658 = P.p[4] * v650;
        const T v659 = v631 + v658;
        const T v663 = z[10] * v645;
        const T v664 = v659 + v663;
        const T v666 = z[11] * v645;
        const T v667 = v655 + v666;
        const T v669 = v23 * v584;
        const T v670 = v669 / v19;
        const T v671 = -v670;
        const T v672 = v577 + v671;
        const T v677 = v18 * v672;
        const T v678 = -v677;
        const T v679 = v19 * v581;
        const T v680 = v678 + v679;
        const T v681 = v16 * v667;
        const T v682 = -v681;
        const T v683 = v17 * v664;
        const T v684 = v682 + v683;
        const T v685 = -v438;
        const T v686 = -v20;
        const T v687 = v436 * v686;
        const T v692 = v21 * v421;
        const T v693 = v420 * v686;
        const T v699 = v413 * v686;
        const T v704 = v403 * v686;
        const T v705 = v439 + v704;
        const T v707 = v21 * v402;
        const T v708 = v687 + v707;
        const T v714 = v393 * v686;
        const T v715 = v692 + v714;
        const T v717 = v397 + v693;
        const T v723 = v21 * v387;
        const T v724 = v699 + v723;
        const T v728 = v18 * v717;
        const T v729 = v705 + v728;
        const T v730 = v16 * v717;
        const T v731 = v18 * v708;
        const T v732 = v715 + v731;
        const T v733 = v17 * v708;
        const T v734 = v730 + v733;
        const T v735 = v20 * v448;
        const T v736 = -v735;
        const T v737 = v21 * v685;
        const T v738 = v736 + v737;
        const T v739 = v18 * v724;
        const T v740 = -v739;
        const T v741 = v19 * v734;
        const T v742 = v740 + v741;
        const T v743 = v16 * v732;
        const T v744 = -v743;
        const T v745 = v17 * v729;
        const T v746 = v744 + v745;
        const T v748 = -cc[3];
        const T v749 = v20 * cc[2];
        const T v750 = v19 * cc[2];
        const T v751 = v21 * cc[1];
        const T v752 = v749 + v751;
        const T v753 = v19 * cc[1];
        const T v754 = v20 * v753;
        const T v755 = -v754;
        const T v756 = v21 * v750;
        const T v757 = v755 + v756;
        const T v758 = v18 * v752;
        const T v759 = -v758;
        const T v760 = v19 * v748;
        const T v761 = v759 + v760;
        const T v767 = v19 * cc[3];
        const T v768 = v16 * cc[3];
        const T v769 = v40 * cc[2];
        const T v770 = v21 * cc[2];
        const T v771 = v17 * cc[2];
        const T v772 = -cc[1];
        const T v773 = v40 * cc[1];
        const T v774 = v771 + v773;
        const T v775 = v758 + v767;
        const T v776 = v16 * v752;
        const T v777 = v20 * v772;
        const T v778 = v770 + v777;
        const T v779 = v17 * v772;
        const T v780 = v769 + v779;
        const T v781 = v20 * v774;
        const T v782 = -v781;
        const T v783 = v21 * v780;
        const T v784 = v782 + v783;
        const T v785 = v18 * v768;
        const T v786 = -v785;
        const T v787 = v19 * v776;
        const T v788 = v786 + v787;
        const T v789 = v16 * v778;
        const T v790 = -v789;
        const T v791 = v17 * v775;
        const T v792 = v790 + v791;
        const T v794 = v32 * P.p[7];
        const T v796 = P.p[9] * v794;
        const T v797 = v331 * v796;
        const T v799 = v331 * v794;
        const T v820 = P.p[9] * v799;
        const T v821 = v797 + v820;
        const T v824 = P.p[7] * v821;
        const T v827 = v32 * v824;
        const T v830 = v17 * cc[3];
        const T v831 = -cc[2];
        const T v832 = v35 * cc[2];
        const T v833 = v21 * v831;
        const T v834 = v16 * v831;
        const T v835 = v35 * cc[1];
        const T v836 = v834 + v835;
        const T v837 = v17 * v752;
        const T v838 = v20 * cc[1];
        const T v839 = v833 + v838;
        const T v840 = v16 * cc[1];
        const T v841 = v832 + v840;
        const T v846 = v20 * v836;
        const T v847 = -v846;
        const T v848 = v21 * v841;
        const T v849 = v847 + v848;
        const T v850 = v18 * v830;
        const T v851 = -v850;
        const T v852 = v19 * v837;
        const T v853 = v851 + v852;
        const T v854 = v16 * v775;
        const T v855 = -v854;
        const T v856 = v17 * v839;
        const T v857 = v855 + v856;
        const T v863 = v136 * v211;
        const T v869 = v141 * v203;
        const T v881 = v16 * cc[4];
        const T v883 = v16 * cc[6];
        const T v886 = v131 * v223;
        const T v906 = v23 * v881;
        const T v907 = v18 * v881;
        const T v908 = v883 + v907;
        const T v910 = v23 * v908;
        const T v911 = v910 / v19;
        const T v912 = -v911;
        const T v913 = v18 * v912;
        const T v914 = -v913;
        const T v915 = v19 * v906;
        const T v916 = v914 + v915;
        const T v917 = v16 * cc[5];
        const T v918 = -v917;
        const T v919 = v362 + v918;
        const T v920 = v17 * cc[4];
        const T v922 = v17 * cc[6];
        const T v943 = v23 * v920;
        const T v944 = v18 * v920;
        const T v945 = v922 + v944;
        const T v947 = v23 * v945;
        const T v948 = v947 / v19;
        const T v949 = -v948;
        const T v950 = v18 * v949;
        const T v951 = -v950;
        const T v952 = v19 * v943;
        const T v953 = v951 + v952;
        const T v954 = -v358;
        const T v955 = v533 + v954;
        const T v960 = v193 * T(2.00000000000000000e+00);
        H[9] += v571;
        H[13] += v684;
        H[14] += v680;
        H[18] += v746;
        H[19] += v742;
        H[20] += v738;
        H[25] += v761;
        H[26] += v757;
        H[31] += v792;
        H[32] += v788;
        H[33] += v784;
        H[39] += v857;
        H[40] += v853;
        H[41] += v849;
        H[44] += v827;
        H[52] += v267;
        H[53] += cc[8];
        H[58] += v919;
        H[59] += v916;
        H[61] += cc[9];
        H[63] += v301;
        H[64] += v869;
        H[69] += v955;
        H[70] += v953;
        H[72] += v282;
        H[73] += cc[7];
        H[75] += v863;
        H[76] += v886;
        H[90] += v960;
        H[104] += v960;
        H[119] += v960;
        H[135] += v960;
        // ---- generated body end ----
    }
};

}  // namespace emi
