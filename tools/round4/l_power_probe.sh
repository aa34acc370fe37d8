#!/bin/bash
# round 4 call L: what power telemetry does the box offer?
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4l; mkdir -p $O
{
which rocm-smi amd-smi 2>&1
timeout 20 rocm-smi --showpower --showclocks --showperflevel 2>&1 | head -40
ls /sys/class/drm/ 2>&1 | head
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h | head -40; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input; do [ -f $h/$f ] && echo "$f=$(cat $h/$f)"; done; done
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -20
timeout 20 amd-smi metric --power --clock 2>&1 | head -40
} > $O/telemetry.txt 2>&1
head -80 $O/telemetry.txt
