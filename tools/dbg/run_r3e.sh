R=$PWD; cd /tmp; export TMPDIR=/tmp
for st in 10 40; do rm -rf /tmp/pc$st; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pc$st -o s -- python $R/bench.py --steps $st --warmup 4 --no-cpu-baseline > /dev/null 2>&1; echo "steps=$st:"; grep -h "copyBuffer\|k_pack" $(find /tmp/pc$st -name "*kernel_stats.csv") | cut -d, -f1-2; done
