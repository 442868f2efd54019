#!/bin/bash
# Socket power and shader clock (rocm-smi) while a command keeps the GPU busy:
#   power_probe.sh "<command>" "<label>" [samples=10] [delay_s=4]
# Prints one JSON line per sample (rocm-smi -P -c -t) taken while the command runs, then the command's own "{" lines.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
echo "=== $2"
eval "$1" > /tmp/power_probe_$$.log 2>&1 &
PID=$!
sleep ${4:-4}     # import torch + warm-up
for i in $(seq 1 ${3:-10}); do
  kill -0 $PID 2>/dev/null || break
  rocm-smi -P -c -t --json 2>/dev/null | tr -d '\n' | cut -c1-900; echo
done
wait $PID
grep "^{" /tmp/power_probe_$$.log | cut -c1-400
rm -f /tmp/power_probe_$$.log
