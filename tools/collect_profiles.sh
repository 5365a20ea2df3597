#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run through gpurun) into gpurun_out/profiles_rNN/.
# usage: bash tools/collect_profiles.sh r04
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. plain bench line (no profiler attached)
python3 $ROOT/bench.py --pyramid > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err || true
# 2. rocprofv3 kernel trace + stats of the same command
# (the default command: single-level headline; --pyramid would launch the same kernels at 8 sizes and blur the averages)
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $ROOT/bench.py > $OUT/bench_${TAG}_under_rocprof.json 2>/dev/null || true
python3 $ROOT/tools/summarize_stats.py /tmp/prof_kt 40 | grep -v "at::native\|rocprim\|rocclr\|compute_cuda" > $OUT/kernel_stats_${TAG}.txt || true
# 3. PMC passes (front-end kernels, 200 frames; separate passes as the guide prescribes)
pmc() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $ROOT/bench.py --frames 200 --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-pyramid > /dev/null 2>&1 || true; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 $ROOT/tools/summarize_pmc.py /tmp/pmc_sq1 /tmp/pmc_sq2 /tmp/pmc_fetch /tmp/pmc_write --traffic-json $OUT/traffic.json 400 > $OUT/pmc_frontend_${TAG}.txt || true
# 3a. orient_rbrief_kernel alone: issue / wait split, LDS, L1 <- L2 requests of its patch gathers
bash $ROOT/tools/pmc_orient.sh $OUT/pmc_orient_${TAG}.txt 200 || true
# 3b. the 8-level pyramid mode (500 frames, 3 passes)
rm -rf /tmp/prof_py && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_py -- python3 $ROOT/tools/pyramid_profile.py 500 > /dev/null 2>&1 || true
python3 $ROOT/tools/summarize_stats.py /tmp/prof_py 40 | grep -v "at::native\|rocprim\|rocclr\|compute_cuda" > $OUT/kernel_stats_pyramid_${TAG}.txt || true
# 4. BA kernels at configs[2]: stats of three linear solves + PMC passes (VALU / LDS / MFMA / L1-L2 / HBM, separate passes)
rm -rf /tmp/prof_ba && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -- python3 $ROOT/tools/ba_profile.py > /dev/null 2>&1 || true
python3 $ROOT/tools/summarize_stats.py /tmp/prof_ba 40 | grep -v "at::native\|rocprim\|rocclr\|compute_cuda" > $OUT/kernel_stats_ba_${TAG}.txt || true
python3 $ROOT/tools/kernel_timeline.py /tmp/prof_ba > $OUT/timeline_ba_${TAG}.txt || true      # the last solve, kernel by kernel
bash $ROOT/tools/pmc_ba.sh $OUT/pmc_ba_${TAG}.txt || true
# 5. VALU issue-rate micro-benchmark (what bounds the FAST kernel) and the f64 matrix / vector rates (BA rooflines)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $ROOT/tools/ubench/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate > $OUT/valu_issue_rates_${TAG}.txt || true
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $ROOT/tools/ubench/mfma_f64_rate.hip -o /tmp/mfma_f64_rate 2>/dev/null && /tmp/mfma_f64_rate > $OUT/mfma_f64_rate_${TAG}.txt || true
/opt/rocm/bin/hipcc -O3 -w --offload-arch=gfx950 $ROOT/tools/ubench/lds_add_f64_rate.hip -o /tmp/lds_add_f64_rate 2>/dev/null && /tmp/lds_add_f64_rate > $OUT/lds_add_f64_rate_${TAG}.txt || true
# 6. set-up of a configs[2] problem: pack, pair lists (torch sort vs csrc/structure.hip) with the kernel durations
rm -rf /tmp/prof_st && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python3 $ROOT/tools/setup_prof.py > $OUT/setup_${TAG}.txt 2>/dev/null || true
python3 $ROOT/tools/summarize_stats.py /tmp/prof_st 60 | grep -i "tiles_\|rs_\|kernel   " >> $OUT/setup_${TAG}.txt || true
# 6b. the landmark elimination alone: one vus_ba_schur call timed with HIP events + kernel stats + PMC (MFMA, LDS, L2, HBM)
python3 $ROOT/tools/schur_ab.py c2 2>/dev/null | grep '^{' > $OUT/schur_${TAG}.json || true
python3 $ROOT/tools/schur_ab.py c4 2>/dev/null | grep '^{' >> $OUT/schur_${TAG}.json || true
bash $ROOT/tools/schur_pmc.sh $OUT/pmc_schur_${TAG}.txt || true
# 7. round 3: the band solve's panel-step modes A/B in one process (2 = launch pairs on two streams, 3 = persistent window
#    kernel), and the first optimize() of a process by phase
python3 $ROOT/tools/band_modes_probe.py 2 3 2>/dev/null | grep '^{' > $OUT/band_modes_${TAG}.txt || true
python3 $ROOT/tools/cold_phases.py 2>/dev/null | grep '^{' > $OUT/cold_phases_${TAG}.txt || true
# 8. where a panel step of the window kernel's critical workgroup goes: s_memtime marks of a -DVUS_TIMING build of the library
CS=$ROOT/visual-underwater-slam_amd/csrc
OFFLOAD=$(make -s -C $CS print-offload)
mkdir -p /tmp/tb && (cd $CS && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC $OFFLOAD -Wno-unused-function -DVUS_TIMING -c ba.hip -o /tmp/tb/ba_t.o 2>/dev/null \
  && /opt/rocm/bin/hipcc -shared -fPIC $OFFLOAD vus_common.o frontend.o /tmp/tb/ba_t.o nav.o pack.o -o /tmp/tb/libvus_timing.so) \
  && VUS_HIP_LIB=/tmp/tb/libvus_timing.so python3 $ROOT/tools/win_timing.py 2>/dev/null | grep '^{' > $OUT/window_step_cycles_${TAG}.txt || true
ls -la $OUT
