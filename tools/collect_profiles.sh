#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics, HBM / SQ counters (separate --pmc passes, as the
# MI355X guide prescribes), the per-phase cycle profile and the bench line of the current build.  Output: gpurun_out/<tag>/
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
sha256sum $ROOT/pypownet_amd/libppn.so > $OUT/libppn.sha256
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --headline-only"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/pmc_sq -o p --output-format csv -- $BENCH > /dev/null 2>&1
# the counters themselves against kernels that move a known number of bytes (tools/ubench/hbm_counter_calib.hip; built in the dev container)
if [ -x $ROOT/build/hbm_counter_calib ]; then
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/calib_write -o p --output-format csv -- $ROOT/build/hbm_counter_calib > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/calib_fetch -o p --output-format csv -- $ROOT/build/hbm_counter_calib > /dev/null 2>&1
  python $ROOT/tools/ubench/hbm_counter_calib.py $OUT/calib_write $OUT/calib_fetch > $OUT/hbm_counter_calib.json 2> $OUT/hbm_counter_calib.err
fi
cd $ROOT
if [ -n "$PPN_COLLECT_PMC_ONLY" ]; then ls $OUT; exit 0; fi      # (a library rebuilt after the long collection: only the passes roofline.traffic is read from)
# every kernel of the default bench run (headline + the other configurations: W = 1 and W = 4 step kernels too)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats_all -o bench_all --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-rollout > $OUT/bench_all_under_rocprof.log 2>&1
cd $ROOT
# (build/libppn_prof.so is built in the development container: python __graft_entry__.py variant prof -DPPN_PROF -- build/ travels)
python tools/profile_phases.py 4096 10 > $OUT/phase_profile_b4096.txt 2>&1
PPN_PROF_ENV=default14 python tools/profile_phases.py 1024 20 > $OUT/phase_profile_default14_b1024.txt 2>&1
PPN_PROF_ENV=default14 python tools/profile_phases.py 16384 10 > $OUT/phase_profile_default14_b16384.txt 2>&1
python tools/profile_phases.py 1024 8 split > $OUT/phase_profile_split_b1024.txt 2>&1
python tools/profile_phases.py 32768 4 > $OUT/phase_profile_b32768.txt 2>&1
PPN_PROF_SOLVER=fdxb python tools/profile_phases.py 4096 10 > $OUT/phase_profile_fdxb_b4096.txt 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
PPN_LAUNCH_ORDER=0 python bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_no_launch_order.json 2>/dev/null
python bench.py --batch 32768 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_b32768.json 2>/dev/null
PPN_BENCH_CHRONICS=2 python bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_2chronics.json 2>/dev/null
PPN_BENCH_CHRONICS=2 python bench.py --batch 32768 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_2chronics_b32768.json 2>/dev/null
tail -1 $OUT/bench.json | cut -c1-400
ls $OUT $OUT/stats | head -30
# afterwards, in the development container: python tools/summarize_pmc.py $TAG  (-> profiles/)
# ---- round 5 extras ------------------------------------------------------------------------------------------------------------
# closed-loop device policy: stepped vs one launch (ppn_rollout_policy)
for b in 4096 8192; do python tools/policy_rate.py $b 60 >> $OUT/policy_rate.txt 2>/dev/null; done
# configs[4] workload: schedule pre-pass off / one wave / four waves, two-capacity stepping on / off (same box)
for pp in 0 64 256 1; do for cfg in "1024 24 tuned" "8192 12 tuned"; do echo "PPN_SCHED_PREPASS=$pp $cfg: $(PPN_SCHED_PREPASS=$pp python tools/split_rate.py $cfg 2>/dev/null | tail -1)"; done; done > $OUT/split_prepass_ab.txt 2>&1
for tc in 1 0; do for cfg in "1024 24 safe" "8192 12 safe"; do echo "PPN_TWO_CAP=$tc $cfg: $(PPN_TWO_CAP=$tc python tools/split_rate.py $cfg 2>/dev/null | tail -1)"; done; done > $OUT/split_two_cap_ab.txt 2>&1
# schedule_build phase by phase: inside the step kernel (one wave), in the pre-pass (four waves / one wave)
PPN_SCHED_PREPASS=0 PPN_TWO_CAP=0 python tools/profile_phases.py 1024 8 split > $OUT/phase_profile_split_b1024_build_in_step_kernel.txt 2>&1
PPN_SCHED_PREPASS=256 PPN_TWO_CAP=0 python tools/profile_phases.py 1024 8 split > $OUT/phase_profile_split_b1024_prepass_4waves.txt 2>&1
PPN_SCHED_PREPASS=64 PPN_TWO_CAP=0 python tools/profile_phases.py 1024 8 split > $OUT/phase_profile_split_b1024_prepass_1wave.txt 2>&1
# soaks: four-word kernels under random actions (>= 10^6 solves per solver), bench workload
python tests/tools/soak_random.py default118 newton 2048 60 6 > $OUT/soak_random_w4_newton.txt 2>&1
python tests/tools/soak_random.py default118 fdxb 2048 60 6 > $OUT/soak_random_w4_fdxb.txt 2>&1
python tests/tools/soak_parity.py 4096 300 20 > $OUT/soak_parity.txt 2>&1
tail -n 2 $OUT/soak_random_w4_newton.txt $OUT/soak_random_w4_fdxb.txt $OUT/soak_parity.txt $OUT/policy_rate.txt
# anatomy of the synchronous headline: which environments end a launch, environments per CU, one environment per CU
python tests/tools/chain_lengths.py 2>&1 | grep -v Warn > $OUT/chain_lengths.txt
for pad in 0 3500 9000 17000 30000; do echo "PPN_LDS_PAD=$pad: $(PPN_LDS_PAD=$pad python tests/tools/lib_compare.py default118 newton 4096 60 default 2>/dev/null | tail -1)"; done > $OUT/occupancy_sweep.txt
python tools/profile_phases.py 256 10 > $OUT/phase_profile_b256_one_env_per_cu.txt 2>&1
# ---- round 6 extras ------------------------------------------------------------------------------------------------------------
# asynchronous session (ppn_send / ppn_recv): rates for server sizes / min_ready, and the anatomy of a served step (profiling build)
python tools/async_rate.py 4096 60 256,1024,2048 0,1536 2>&1 | grep -v amdgpu.ids > $OUT/async_rate.txt
PPN_ASYNC_ANATOMY=1 PPN_DEBUG_LIB=build/libppn_prof.so python tools/dev/async_debug.py 4096 40 devact+idsdev+mr1024 rollout 2>&1 | grep -v amdgpu.ids > $OUT/async_anatomy.txt
python tools/search_rate.py > $OUT/search_rate.txt 2>&1
# the 100k-solve random-action soak of the -m gpu suite is the short form of the two soaks above
# closed-loop rollout kernel (XCD-affine hand-out) against the stepped form at the headline's size, and its rates
python tests/tools/soak_policy_rollout.py 4096 40 3 2>&1 | grep -v amdgpu.ids | tail -7 > $OUT/soak_policy_rollout.txt
# the restart memo against the oracle over 300 steps (deferred restart: the memo's apply pass in front of every step)
PPN_RESTART_MEMO=1 PPN_SOAK_AUTO_RESET=2 python tests/tools/soak_parity.py 4096 300 20 2>&1 | tail -1 > $OUT/soak_parity_restart_memo.txt
