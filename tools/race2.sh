# two concurrent processes on one GPU, each repeating the same training step without weight updates (tools/wgrad_race_probe.py):
# run-to-run deviations under contention = a stream-ordering bug somewhere in the step.  usage: bash tools/race2.sh [steps]
mkdir -p gpurun_out/flaky
N=${1:-30}
run2() {
  label=$1; shift
  echo "== $label"
  ( env "$@" timeout 600 python tools/wgrad_race_probe.py ${MODE:-f32} $N 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | sed "s/^/A: /" ) > gpurun_out/flaky/_a.txt &
  ( env "$@" timeout 600 python tools/wgrad_race_probe.py ${MODE:-f32} $N 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-200 | sed "s/^/B: /" ) > gpurun_out/flaky/_b.txt &
  wait
  head -4 gpurun_out/flaky/_a.txt; tail -1 gpurun_out/flaky/_a.txt; head -4 gpurun_out/flaky/_b.txt; tail -1 gpurun_out/flaky/_b.txt
}
for rep in 1 2; do
run2 "default $rep" X=1
run2 "serialize-kernel $rep" AMD_SERIALIZE_KERNEL=3
done
