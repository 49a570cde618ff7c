bash tools/prof_timeline.sh; cp gpurun_out/timeline.txt gpurun_out/r3b_timeline.txt
python tools/rocpd_stats.py $(find /tmp/p2 -name "*.db" | head -1) > gpurun_out/r3b_stats.txt
bash tools/prof_serial.sh; cp gpurun_out/stats_serial.txt gpurun_out/r3b_stats_serial.txt
