#!/bin/bash
# which sysfs files give the shader clock / socket power on this box (bench.py samples them during the timed regions)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/vendor $d/device 2>/dev/null | tr '\n' ' '; echo
  for f in $d/hwmon/hwmon*/freq*_input $d/hwmon/hwmon*/freq*_label $d/hwmon/hwmon*/power*_average $d/hwmon/hwmon*/power*_input $d/hwmon/hwmon*/power*_label $d/pp_dpm_sclk $d/gpu_busy_percent; do
    [ -e $f ] && echo "$f: $(cat $f 2>&1 | tr '\n' '|')"
  done
done
time rocm-smi --showclocks --showpower --csv 2>&1 | head -5
