mkdir -p gpurun_out/flaky
for mode in f32 bf16; do
  echo "== $mode, two processes"
  ( timeout 600 python tools/component_race_probe.py $mode ${1:-60} $2 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | sed "s/^/A: /" ) > gpurun_out/flaky/_a.txt &
  ( timeout 600 python tools/component_race_probe.py $mode ${1:-60} $2 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | sed "s/^/B: /" ) > gpurun_out/flaky/_b.txt &
  wait
  tail -12 gpurun_out/flaky/_a.txt; tail -12 gpurun_out/flaky/_b.txt
done
echo "== f32, ONE process"
timeout 600 python tools/component_race_probe.py f32 ${1:-60} $2 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | tail -5
