#!/bin/bash
# tools/rss_watch.sh <pid> <logfile>: records the peak resident set size (VmHWM) of a process every 20 s until it exits
pid=$1; log=$2
while [ -d /proc/$pid ]; do
  grep -E 'VmHWM|VmRSS' /proc/$pid/status 2>/dev/null | tr '\n' ' ' > "$log.tmp" && { date +%T | tr '\n' ' '; cat "$log.tmp"; echo; } >> "$log"
  sleep 20
done
rm -f "$log.tmp"
