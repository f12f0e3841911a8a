#!/bin/bash
# round 4, GPU call 6: rocprofv3 evidence of the final build, part 1 (c3, c3 deformable, c2)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for args in "c3 bf16 r04" "c3 bf16 r04 deformable" "c2 f32_split3 r04"; do
  timeout -k 10 360 bash tools/profile_config.sh $args > gpurun_out/r04_profile_$(echo $args | tr ' ' '_').log 2>&1 || { echo "profile $args failed"; tail -5 gpurun_out/r04_profile_$(echo $args | tr ' ' '_').log; exit 1; }
  tail -2 gpurun_out/r04_profile_$(echo $args | tr ' ' '_').log
done
ls gpurun_out/profiles_out | head -40
