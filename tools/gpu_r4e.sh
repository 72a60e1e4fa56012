#!/bin/bash
# round 4, fifth GPU call: parity of the new default (SW 2 / 16-deep tiles at 128..255 instances), Newton-step piece times single vs
# batched, mid-range sweep 8- vs 16-deep, Monte-Carlo with one rendezvous per mesh size, config 5 with the node kernel beside the ring
# kernel, logs of two stragglers
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py tests/test_gpu_solve.py -m gpu -q -x > gpurun_out/pytest_r4e.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_r4e.log
rm -f gpurun_out/kkt_times.jsonl
timeout -k 10 200 python tools/kkt_times.py --nodes 1024 --batch 1,4,16,32 --lowrank 770 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/kkt_times.py --nodes 257 --batch 1,16,64 --lowrank 150 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/kkt_times.py --nodes 65 --batch 1,16,64 --lowrank 40 2>&1 | grep -v amdgpu.ids
rm -f gpurun_out/mid_sweep_r4e.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 128,224,320,384,448,576,640,768,896 --rounds 5 --out gpurun_out/mid_sweep_r4e.jsonl \
  --forms default,bk16 2>&1 | grep -v amdgpu.ids
: > gpurun_out/mc_r4e.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads groups flush_us
  EMI_MC_FLUSH_US=$6 EMI_MC_BATCH=$5 EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4e_$2_t$4_g$5_f$6.log 2>&1
  echo "mc $* rc=$?"; grep -E "^batcher" gpurun_out/mc_r4e_$2_t$4_g$5_f$6.log | head -4 | cut -c1-200; tail -1 gpurun_out/mc_r4e_$2_t$4_g$5_f$6.log | tee -a gpurun_out/mc_r4e.jsonl
}
run 64 1023 20 8 0 0
run 64 1023 20 32 1 20000
run 64 1023 20 64 1 20000
run 64 1023 20 64 1 100000
run 64 256 10 8 0 0
run 64 256 10 32 1 20000
run 64 256 10 64 1 20000
run 64 128 10 64 1 20000
python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('c5 default', j['ms_per_step'], j['roofline']['avg_ms'], j['roofline']['kernel'])"
EMI_OVERLAP_MODE=2 EMI_F32_RING_WGS=1 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('c5 two streams ring 1 wg/cu', j['ms_per_step'], j['roofline']['avg_ms'], j['roofline']['kernel'])"
EMI_OVERLAP_MODE=2 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('c5 two streams ring 2 wg/cu', j['ms_per_step'], j['roofline']['avg_ms'], j['roofline']['kernel'])"
for s in 42 17; do
  EMI_MC_GATHER=0 EMI_MC_ONLY=$s EMI_MC_PRINT_LEVEL=5 timeout -k 10 120 etol_amd/lib/etol_mi355x_montecarlo 64 1023 20 1 > gpurun_out/straggler_$s.log 2>&1
  grep -E "mesh sequencing|^scenario|stagnation|ladder|warm start" gpurun_out/straggler_$s.log | cut -c1-200
done
