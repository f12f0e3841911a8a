#!/bin/bash
# Poll board power and shader clock from sysfs while a bench configuration runs: tools/power_poll.sh c3 [steps] > log
# (is the bf16 path power / DVFS limited?  MI355X_MICROARCH.md, DVFS give-back)
CFG=${1:-c3}; STEPS=${2:-150}
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
DEV=$(dirname $(dirname $HW))
echo "hwmon $HW dev $DEV"; ls $HW | tr '\n' ' '; echo
cat $HW/power1_cap 2>/dev/null | sed 's/^/power1_cap uW /'
python bench.py --config $CFG --cpu-baseline off --profile-steps 0 --other-configs off --steps $STEPS --warmup 10 > gpurun_out/power_poll_bench_$CFG.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  P=$(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null)
  F=$(cat $HW/freq1_input 2>/dev/null)
  S=$(grep '\*' $DEV/pp_dpm_sclk 2>/dev/null | tr -d '\n')
  T=$(cat $HW/temp1_input 2>/dev/null)
  echo "$(date +%s.%N) power_uW=$P freq1_Hz=$F sclk=[$S] temp_mC=$T"
  sleep 0.25
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/power_poll_bench_$CFG.json')); print('bench', d['value'], d['ms_per_step'])"
