#!/bin/bash
# One GPU-box visit (gpurun): named steps, each under its own timeout, logs under gpurun_out/<tag>_*.
#   tools/gpu_visit.sh <tag> <step> [<step> ...]
# steps: see the case labels below (tests, smoke, bench lines, kernel traces, probes)
set -u
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
prof_db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
for STEP in "$@"; do
  echo "=== $STEP"
  case $STEP in
    fused_tests)
      timeout 600 python -m pytest tests/test_gpu_fused_step.py tests/test_end_to_end_bench_config.py -x -q > gpurun_out/${TAG}_fused_tests.log 2>&1
      echo "rc=$?"; tail -5 gpurun_out/${TAG}_fused_tests.log ;;
    rows_tests)
      timeout 300 python -m pytest tests/test_gpu_fused_step.py -x -q -k "(bf16_gradients and (rows or cols32)) or tile_image or rollout_step_bf16" > gpurun_out/${TAG}_rows_tests.log 2>&1
      echo "rc=$?"; tail -3 gpurun_out/${TAG}_rows_tests.log ;;
    xgmi_tests)
      timeout 600 python -m pytest tests/test_distributed.py -x -q -m gpu -k "in_process_group" > gpurun_out/${TAG}_xgmi_tests.log 2>&1
      echo "rc=$?"; tail -4 gpurun_out/${TAG}_xgmi_tests.log ;;
    valuefree_tests)
      timeout 600 python -m pytest tests/test_end_to_end.py -x -q -m gpu -k "value_free or embodied_grpo" > gpurun_out/${TAG}_valuefree_tests.log 2>&1
      echo "rc=$?"; tail -15 gpurun_out/${TAG}_valuefree_tests.log | cut -c1-300 ;;
    dist_tests)
      timeout 1500 python -m pytest tests/test_distributed.py -x -q -m gpu -k "eight_ranks or bench_self_launches" > gpurun_out/${TAG}_dist_tests.log 2>&1
      echo "rc=$?"; tail -12 gpurun_out/${TAG}_dist_tests.log | cut -c1-400 ;;
    bench8)
      ( unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT; export RLX_BENCH_ALLOW_SHARED_GPU=1 RLX_DIST_BACKEND=gloo RLX_XGMI_TIMEOUT_MS=120000
        T0=$SECONDS; timeout 1700 python bench.py --gpus 8 --steps 3 --warmup 1 --no-roofline --launch-timeout 1600 --pair-timeout 500 > gpurun_out/${TAG}_bench8.json 2> gpurun_out/${TAG}_bench8.err
        echo "rc=$? wall=$((SECONDS - T0)) s"; tail -1 gpurun_out/${TAG}_bench8.json | cut -c1-1500 ) ;;
    loss_tests)
      timeout 600 python -m pytest tests/test_gpu_losses.py -x -q -m gpu -k "chunk_level_reward_with_token_level" > gpurun_out/${TAG}_loss_tests.log 2>&1
      echo "rc=$?"; tail -25 gpurun_out/${TAG}_loss_tests.log | cut -c1-300 ;;
    all_tests)
      timeout 1500 python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
      echo "rc=$?"; tail -42 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-180 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    async_tests)
      timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_losses.py tests/test_end_to_end.py -q -x -m gpu -k "decoupled or deferred or async" 2>&1 | tail -25 ;;
    async_dist_tests)
      timeout 1200 python -m pytest tests/test_distributed.py -q -x -m gpu -k "async_learner or two_ranks_match or xgmi_clip_adamw" 2>&1 | tail -25 | cut -c1-400 ;;
    run_ahead_tests)
      timeout 900 python -m pytest tests/test_end_to_end.py -q -x -m gpu -k "run_ahead or async" 2>&1 | tail -15 | cut -c1-300 ;;
    bench_variants)
      timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_variants.json 2> gpurun_out/${TAG}_bench_variants.err
      echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_variants.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sync', d['ms_per_step']); [print(v.get('variant','')[:50], v.get('ms_per_step'), v.get('error')) for v in d.get('variants',[])]" ;;
    reasoning_loop)
      timeout 300 python -m pytest tests/test_gpu_token_path.py tests/test_gpu_reasoning_loop.py -x -q 2>&1 | tail -3
      timeout 600 python tools/bench_reasoning_loop.py > gpurun_out/${TAG}_reasoning_loop.json 2> gpurun_out/${TAG}_reasoning_loop.err; echo "rc=$?"; tail -3 gpurun_out/${TAG}_reasoning_loop.err; cat gpurun_out/${TAG}_reasoning_loop.json | cut -c1-900
      timeout 600 python tools/bench_reasoning_loop.py --no-inplace-grad 2>/dev/null | tee -a gpurun_out/${TAG}_reasoning_loop.json | cut -c300-800
      rm -rf gpurun_out/prof_reason
      timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_reason -o reason -- python tools/bench_reasoning_loop.py --iters 2 > /dev/null 2>&1
      python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_reason)" > gpurun_out/${TAG}_reasoning_loop_kernels.txt 2>&1; head -16 gpurun_out/${TAG}_reasoning_loop_kernels.txt | cut -c1-170; rm -rf gpurun_out/prof_reason ;;
    async_prof)
      rm -rf gpurun_out/prof_async
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_async -o bench -- python bench.py --learner async --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${TAG}_async_prof.log 2>&1
      echo "rc=$?"; tail -1 gpurun_out/${TAG}_async_prof.log | cut -c1-200; python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_async)" > gpurun_out/${TAG}_async_kernels.txt 2>&1; head -16 gpurun_out/${TAG}_async_kernels.txt | cut -c1-170; rm -rf gpurun_out/prof_async ;;
    rows_probe)
      timeout 300 python tools/fused_rows_probe.py > gpurun_out/${TAG}_rows_probe.txt 2>&1; echo "rc=$?"; cat gpurun_out/${TAG}_rows_probe.txt ;;
    rows_dev)
      timeout 300 python tools/fused_rows_probe.py --dev --sizes 8192,1024 --iters 100 > gpurun_out/${TAG}_rows_dev.txt 2>&1; echo "rc=$?"; cat gpurun_out/${TAG}_rows_dev.txt ;;
    rows_probe_prof)
      rm -rf gpurun_out/prof_rows
      timeout 400 rocprofv3 --kernel-trace -d gpurun_out/prof_rows -o rows -- python tools/fused_rows_probe.py --iters 100 > gpurun_out/${TAG}_rows_probe_prof.log 2>&1
      echo "rc=$?"; python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_rows)" rlx --by-grid > gpurun_out/${TAG}_rows_probe_kernels.txt 2>&1
      cat gpurun_out/${TAG}_rows_probe_kernels.txt; rm -rf gpurun_out/prof_rows ;;
    share_prof)
      for ROWS in 1 0; do for SH in 1 8; do
        rm -rf gpurun_out/prof_share
        RLX_FUSED_ROWS=$ROWS timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_share -o share -- python tools/per_rank_share.py --share $SH --steps 20 > gpurun_out/${TAG}_share${SH}_rows${ROWS}.log 2>&1
        echo "rows=$ROWS share=$SH rc=$?"; grep "ms / iteration" gpurun_out/${TAG}_share${SH}_rows${ROWS}.log
        python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_share)" rlx > gpurun_out/${TAG}_share${SH}_rows${ROWS}_kernels.txt 2>&1
        head -9 gpurun_out/${TAG}_share${SH}_rows${ROWS}_kernels.txt; rm -rf gpurun_out/prof_share
        RLX_FUSED_ROWS=$ROWS timeout 120 python tools/per_rank_share.py --share $SH --steps 50 2>/dev/null | grep "ms / iteration" | sed "s/^/untraced rows=$ROWS: /" | tee -a gpurun_out/${TAG}_share${SH}_rows${ROWS}.log
      done; done ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300 ;;
    bench_quick)
      timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${TAG}_bench_quick.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_quick.log | cut -c1-400 ;;
    bench_ab)
      for ROWS in 0 1 0 1; do
        RLX_FUSED_ROWS=$ROWS timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${TAG}_bench_rows${ROWS}.log 2>&1
        echo "rows=$ROWS rc=$? $(tail -1 gpurun_out/${TAG}_bench_rows${ROWS}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
      done ;;
    dw_ab)
      for RING in 0 1 0 1; do
        RLX_DW_RING=$RING timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${TAG}_bench_ring${RING}.log 2>&1
        echo "ring=$RING rc=$? $(tail -1 gpurun_out/${TAG}_bench_ring${RING}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
      done
      timeout 120 python tools/phase_times.py 8192 2>/dev/null | grep "dw bf16" | tail -2 ;;
    dw_slabs)
      for CFG in ${DW_CFGS:-0:16 1:16 1:12 1:11 1:10 1:9 1:8 0:10}; do
        set -- ${CFG//:/ }
        rm -rf gpurun_out/prof_dw
        export RLX_DW_RING_NBUF=${3:-4}
        RLX_DW_RING=$1 RLX_DW_SLABS=$2 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_dw -o dw -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${TAG}_dw_prof.log 2>&1
        echo "ring=$1 slabs=$2 nbuf=${3:-4}: $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_dw)" rlx | grep "dw_bf16\|grad_reduce\|clip_adamw\|fused_bf16" | awk '{printf "%s %s | ", substr($2,6,22), $(NF-7)}')"
        RLX_DW_RING=$1 RLX_DW_SLABS=$2 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-roofline --no-variants 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', d['ms_per_step'])"
      done; rm -rf gpurun_out/prof_dw ;;
    bench_prof)
      rm -rf gpurun_out/prof_bench
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-variants --no-token-tier --no-extras --no-roofline > gpurun_out/${TAG}_bench_prof.log 2>&1
      echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_prof.log | cut -c1-200; python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_bench)" > gpurun_out/${TAG}_bench_kernels.txt 2>&1; head -14 gpurun_out/${TAG}_bench_kernels.txt; rm -rf gpurun_out/prof_bench ;;
    r5_tests)
      timeout 900 python -m pytest tests/test_end_to_end.py tests/test_distributed.py tests/test_gpu_token_path.py -q -m gpu -k "value_free or one_rank_drives or categorical" > gpurun_out/${TAG}_r5_tests.log 2>&1
      echo "rc=$?"; tail -30 gpurun_out/${TAG}_r5_tests.log | cut -c1-400 ;;
    overlap_probe)
      timeout 900 python tools/pipeline_overlap_probe.py --steps ${OVERLAP_STEPS:-60} --rounds ${OVERLAP_ROUNDS:-2} > gpurun_out/${TAG}_overlap_probe.txt 2> gpurun_out/${TAG}_overlap_probe.err
      echo "rc=$?"; tail -14 gpurun_out/${TAG}_overlap_probe.txt; tail -3 gpurun_out/${TAG}_overlap_probe.err ;;
    overlap_trace)
      for OV in 1 0; do
        rm -rf gpurun_out/prof_ov
        EXTRA=""; [ $OV = 1 ] && EXTRA="--overlap"
        timeout 200 rocprofv3 --kernel-trace -d gpurun_out/prof_ov -o ov -- python bench.py --pipeline --rollout-epochs 4 $EXTRA --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_overlap_trace_${OV}.log 2>&1
        echo "overlap=$OV rc=$?"; tail -1 gpurun_out/${TAG}_overlap_trace_${OV}.log | cut -c1-160
        python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_ov)" --streams > gpurun_out/${TAG}_overlap_streams_${OV}.txt 2>&1; cat gpurun_out/${TAG}_overlap_streams_${OV}.txt | cut -c1-200
      done; rm -rf gpurun_out/prof_ov ;;
    exchange_self)
      timeout 300 python bench.py --exchange self > gpurun_out/${TAG}_exchange_self.json 2> gpurun_out/${TAG}_exchange_self.err; echo "rc=$?"; cat gpurun_out/${TAG}_exchange_self.json | cut -c1-1500; tail -3 gpurun_out/${TAG}_exchange_self.err ;;
    f32_tests)
      timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_policy.py tests/test_end_to_end_bench_config.py tests/test_end_to_end.py -q -m gpu -x > gpurun_out/${TAG}_f32_tests.log 2>&1
      echo "rc=$?"; tail -40 gpurun_out/${TAG}_f32_tests.log | cut -c1-300 ;;
    bench_f32_prof)
      rm -rf gpurun_out/prof_f32
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f32 -o bench -- python bench.py --precision 32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_bench_f32_prof.log 2>&1
      echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_f32_prof.log | cut -c1-200; python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_f32)" > gpurun_out/${TAG}_bench_kernels_f32.txt 2>&1; head -14 gpurun_out/${TAG}_bench_kernels_f32.txt; rm -rf gpurun_out/prof_f32
      timeout 300 python bench.py --precision 32 --steps 50 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_bench_f32.json 2>gpurun_out/${TAG}_bench_f32.err; echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_f32.json | cut -c1-250 ;;
    f32_pd_sweep)
      for PD in ${F32_PDS:-2 3 1}; do
        rm -rf gpurun_out/prof_f32
        RLX_F32X_PD=$PD timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_f32 -o bench -- python bench.py --precision 32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_f32_pd.log 2>&1
        echo "PD=$PD rc=$? $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_f32)" rlx | grep "f32x\|grad_reduce\|clip_adamw" | awk '{printf "%s %s | ", substr($0,1,40), $(NF-7)}')"
        RLX_F32X_PD=$PD timeout 300 python bench.py --precision 32 --steps 40 --no-cpu-baseline --no-roofline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms', d['ms_per_step'])"
      done; rm -rf gpurun_out/prof_f32 ;;
    f32_rt_ab)
      for RTV in ${F32_RTS:-4 2 4 2}; do
        rm -rf gpurun_out/prof_f32
        RLX_F32X_RT=$RTV timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_f32 -o bench -- python bench.py --precision 32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_f32_rt.log 2>&1
        echo "RT=$RTV rc=$? $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_f32)" rlx | grep "f32x\|grad_reduce\|clip_adamw" | awk '{printf "%s %s | ", substr($0,1,40), $(NF-7)}')"
        RLX_F32X_RT=$RTV timeout 300 python bench.py --precision 32 --steps 40 --no-cpu-baseline --no-roofline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms', d['ms_per_step'])"
      done; rm -rf gpurun_out/prof_f32 ;;
    f32_quick_tests)
      timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_end_to_end_bench_config.py -q -m gpu -x -k "not bf16" > gpurun_out/${TAG}_f32_quick_tests.log 2>&1
      echo "rc=$?"; tail -15 gpurun_out/${TAG}_f32_quick_tests.log | cut -c1-300 ;;
    f32_stamps)
      for RTV in ${F32_RTS:-4 2}; do
        echo "--- RLX_F32X_RT=$RTV"; RLX_F32X_RT=$RTV timeout 200 python tools/phase_times.py 8192 2>&1 | grep -A1 "ppo_step_fused M" | tail -4
      done ;;
    f32_lib_ab)
      for LT in ${F32_LIBS:-a b c main a b c main}; do
        export RLX_LIB_TAG=$LT; [ $LT = main ] && unset RLX_LIB_TAG
        rm -rf gpurun_out/prof_f32
        RLX_F32X_RT=${F32_RT:-2} timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_f32 -o bench -- python bench.py --precision 32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_f32_lib.log 2>&1
        echo "lib=$LT rc=$? $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_f32)" rlx | grep "f32x\|grad_reduce\|clip_adamw" | awk '{printf "%s %s | ", substr($0,1,34), $(NF-7)}')"
      done; unset RLX_LIB_TAG; rm -rf gpurun_out/prof_f32 ;;
    dev_variant_tests)
      RLX_LIB_TAG=dev timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_advantages.py tests/test_gpu_token_path.py tests/test_gpu_weight_patch.py -q -m gpu -k "bf16_gradients or handoff or lookback or zplane_codec_large or variants_vs_oracle" > gpurun_out/${TAG}_dev_variant_tests.log 2>&1
      echo "rc=$?"; tail -6 gpurun_out/${TAG}_dev_variant_tests.log | cut -c1-300 ;;
    pack_tests)
      timeout 900 python -m pytest tests/test_gpu_token_path.py tests/test_gpu_reasoning_loop.py -q -m gpu -k "packed or run_training_matches" > gpurun_out/${TAG}_pack_tests.log 2>&1
      echo "rc=$?"; tail -25 gpurun_out/${TAG}_pack_tests.log | cut -c1-300 ;;
    f32_rollout_pd)
      for PD in 2 4 8 2 4 8; do
        rm -rf gpurun_out/prof_f32
        RLX_LIB_TAG=dev RLX_F32X_ROLLOUT_PD=$PD timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_f32 -o bench -- python bench.py --precision 32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_f32_pd.log 2>&1
        echo "rollout PD=$PD rc=$? $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_f32)" rlx | grep "rollout_step" | awk '{printf "%s %s | ", substr($0,1,40), $(NF-7)}')"
      done; rm -rf gpurun_out/prof_f32 ;;
    bf16_stamps)
      RLX_LIB_TAG=dev timeout 200 python tools/phase_times.py 8192 2>&1 | grep -A1 "ppo_step_fused_bf16\|rollout_step B" | tail -12 ;;
    adamw_one_tests)
      timeout 600 python -m pytest tests/test_gpu_losses.py -x -q -m gpu -k "clip_adamw or one_launch" > gpurun_out/${TAG}_adamw_one_tests.log 2>&1
      echo "rc=$?"; tail -15 gpurun_out/${TAG}_adamw_one_tests.log | cut -c1-300 ;;
    adamw_one_ab)
      for ONE in 0 1 0 1; do
        rm -rf gpurun_out/prof_one
        RLX_ADAMW_ONE_LAUNCH=$ONE timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_one -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-variants --no-extras > gpurun_out/${TAG}_one_prof.log 2>&1
        echo "one_launch=$ONE: $(python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_one)" rlx | grep "dw_bf16\|grad_reduce\|clip_adamw\|fused_bf16" | awk '{printf "%s %s | ", substr($2,6,26), $(NF-7)}')"
        RLX_ADAMW_ONE_LAUNCH=$ONE timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', d['ms_per_step'], d.get('ms_per_step_windows'))"
      done
      for ONE in 0 1; do
        RLX_ADAMW_ONE_LAUNCH=$ONE timeout 300 python bench.py --precision 32 --steps 50 --no-cpu-baseline --no-roofline --no-variants --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   f32 bench one_launch=$ONE', d['ms_per_step'])"
      done; rm -rf gpurun_out/prof_one ;;
    bench_prof_roofline)
      rm -rf gpurun_out/prof_bench
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-variants --no-token-tier --no-extras > gpurun_out/${TAG}_bench_prof_roofline.log 2>&1
      echo "rc=$?"; tail -1 gpurun_out/${TAG}_bench_prof_roofline.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
      python tools/rocpd_stats.py "$(prof_db gpurun_out/prof_bench)" > gpurun_out/${TAG}_bench_kernels_with_roofline_probe.txt 2>&1; grep -n "gae_scan\|kernel  " gpurun_out/${TAG}_bench_kernels_with_roofline_probe.txt | head -6; rm -rf gpurun_out/prof_bench ;;
    *) echo "unknown step $STEP" ;;
  esac
done
