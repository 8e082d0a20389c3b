#!/bin/bash
# Samples rocm-smi power / sclk while a command runs (GPU box):  tools/probes/power_watch.sh <command...>
(for i in $(seq 1 400); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr "\n" " "; echo; sleep 0.2; done > /tmp/smi_watch.log) &
SMI=$!
"$@"
kill $SMI 2>/dev/null
echo "-- rocm-smi samples during the command (count, value):"
sort /tmp/smi_watch.log | uniq -c | sort -k3n | tail -25
