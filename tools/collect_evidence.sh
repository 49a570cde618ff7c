#!/bin/bash
# One gpurun call: kernel-trace stats, the PMC passes (each in its own run, as MI355X_MICROARCH.md prescribes), the profile json
# bench.py quotes as the static half of its roofline object, then the final bench line and the HIP-event table.
# usage (on the GPU box, from the repo root): bash tools/collect_evidence.sh r06_c
set -u
TAG=${1:-r06_c}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shapes --regions 1"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $BENCH > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $BENCH > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $BENCH > /dev/null 2> $O/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o m -- $BENCH > /dev/null 2> $O/mfma.err
cd $R
SDB=$(find $O/stats -name "*.db" | head -1); FDB=$(find $O/fetch -name "*.db" | head -1); WDB=$(find $O/write -name "*.db" | head -1)
python tools/rocpd_stats.py $SDB > $O/${TAG}_kernel_stats.txt
python tools/rocpd_timeline.py $SDB 15 > $O/${TAG}_timeline.txt
python tools/rocpd_pmc.py $FDB > $O/${TAG}_pmc_fetch_size.txt
python tools/rocpd_pmc.py $WDB > $O/${TAG}_pmc_write_size.txt
python tools/rocpd_mfma.py $(find $O/mfma -name "*.db" | head -1) > $O/${TAG}_pmc_mfma_busy.txt
python tools/make_profile_json.py $SDB $FDB $WDB $O/${TAG}_profile.json $TAG
cp $O/${TAG}_profile.json profiles/r06_profile.json
python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --profile-all 2> $O/${TAG}_hip_event_table.txt > /dev/null
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --dtype bf16 > $O/${TAG}_bench_bf16_mode.json 2>/dev/null
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --predictor rnn --batch 16 > $O/${TAG}_bench_rnn_b16_configs0.json 2>/dev/null
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --predictor rnn --batch 64 > $O/${TAG}_bench_rnn_b64.json 2>/dev/null
bash tools/bench_shapes.sh > $O/${TAG}_bench_shapes.txt 2>/dev/null
# round 6: the step's critical-path ledger from HIP events (no profiler attached), batch lengths off the 32-row tile, the row-tile query launches as A/B
python tools/critical_path.py --out $O/${TAG}_critical_path.txt > /dev/null 2> $O/critical_path.err
python tools/dbg/r06_ragged.py 128 117 100 96 > $O/${TAG}_ragged.txt 2>/dev/null
bash tools/dbg/r05_ab_env.sh 3 "VSL_QUERY_FUSED=0" "VSL_LOSS_INLINE=0" "VSL_CQ_FOLD=0" "VSL_FUSED_TAIL=1" - > $O/${TAG}_ab_switches.txt 2>&1
VSL_MULTI_STREAM=0 python tools/critical_path.py --out $O/${TAG}_single_stream_ledger.txt > /dev/null 2>&1
VSL_MULTI_STREAM=0 python tools/dbg/r06_attn_long.py 2>/dev/null | grep drop > $O/${TAG}_attn_long.txt
# per-kernel stats of the other BASELINE configs (configs[0]: the rnn head, with a one-step timeline; configs[2..4]: per-GPU shapes)
cd /tmp
for cfg in "0 --predictor rnn --batch 16" "2 --batch 32 --T 256 --dv 4096" "3 --batch 32 --T 256" "4 --batch 16 --T 1024"; do
  set -- $cfg; n=$1; shift
  rm -rf $O/c$n
  rocprofv3 --kernel-trace --stats -d $O/c$n -o s -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" > /dev/null 2> $O/c$n.err
  python $R/tools/rocpd_stats.py $(find $O/c$n -name "*.db" | head -1) > $O/${TAG}_configs${n}_kernel_stats.txt
  if [ $n = 0 ]; then python $R/tools/rocpd_timeline.py $(find $O/c$n -name "*.db" | head -1) 8 > $O/${TAG}_configs0_timeline.txt; fi
  rm -rf $O/c$n
done
cd $R
rm -rf $O/stats $O/fetch $O/write $O/mfma
tail -c 2500 $O/${TAG}_bench.json
