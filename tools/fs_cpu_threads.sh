#!/bin/bash
# Which host threads burn CPU during strict --fs passes: thread CPU times of the probe process sampled while it runs
# (ps -L: cumulative utime+stime per thread, in clock ticks via /proc).   gpurun -- 'bash tools/fs_cpu_threads.sh [env...]'
cd $GRAFT_REPO_ROOT
env "$@" python3 tools/fs_strict_probe.py --steps 250 > /tmp/fs_cpu_probe.log 2>&1 &
PID=$!
sleep 7
snap() { for t in /proc/$PID/task/*; do echo "$(basename $t) $(awk '{print $14+$15}' $t/stat 2>/dev/null) $(cat $t/comm 2>/dev/null)"; done | sort -k1 -n; }
snap > /tmp/s1; sleep 2; snap > /tmp/s2
join /tmp/s1 /tmp/s2 | awk '{d=$4-$2; if (d>0) print d/2.0, "ticks/s", $1, $3}' | sort -nr | head -25
echo "threads: $(ls /proc/$PID/task | wc -l)"
wait $PID
tail -1 /tmp/fs_cpu_probe.log | cut -c1-80
