timeout 2400 bash tools/collect_evidence.sh r03_b > gpurun_out/evid.log 2>&1
tail -c 600 gpurun_out/evid.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
for st in 10 40; do rm -rf /tmp/pc$st; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc$st -o s -- python $R/bench.py --steps $st --warmup 4 --no-cpu-baseline > /dev/null 2>&1; echo "steps=$st:"; grep -h "copyBuffer\|k_pack" $(find /tmp/pc$st -name "*kernel_stats.csv") | cut -d, -f1-2; done 2>&1 | tee $R/gpurun_out/copybuffer_count.txt
