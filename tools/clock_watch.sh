#!/bin/bash
# run a GPU command in the background and sample the shader clock / power with rocm-smi while it runs
# usage: tools/clock_watch.sh <label> <command...>
label=$1; shift
"$@" > /tmp/cw_$label.out 2>&1 &
pid=$!
sleep 4
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Average Graphics\|Socket" | tr '\n' ' '
  echo
  sleep 1
done
wait $pid
echo "-- $label output:"; tail -3 /tmp/cw_$label.out
