a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d" " -f2); u=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2)
"$@"
b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d" " -f2); v=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2)
echo "   throttled periods: $((b-a)), cpu seconds: $(( (v-u)/1000000 ))"
