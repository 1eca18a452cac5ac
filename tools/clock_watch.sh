#!/bin/bash
# samples the shader clock / power with rocm-smi while a command runs:  tools/clock_watch.sh <logfile> <command...>
log=$1; shift
"$@" > "$log.cmd" 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Power" | tr -s ' ' | tr '\n' ';' >> "$log"
  echo >> "$log"
  sleep 2
done
wait $pid
