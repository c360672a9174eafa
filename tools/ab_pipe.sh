# A/B of the software-pipelined 1x1 kernel inside the headline step: bench.py with MRCNN_CONV_PIPE / MRCNN_PIPE_WGS settings, same box
set -u
mkdir -p gpurun_out/r04e
B="--alt-precision none --alt-config5 0 --cpu-images 0 --measure-traffic 0 --in-flight 1 --steps 10 --reps 2"
for cfg in "0:2" "-1:2" "2:2" "2:3" "4:2" "6:2" "6:3" "3:2" "0:2"; do
  pipe=${cfg%%:*}; wgs=${cfg##*:}
  MRCNN_CONV_PIPE=$pipe MRCNN_PIPE_WGS=$wgs timeout -k 10 300 python bench.py $B > gpurun_out/r04e/b.json 2> gpurun_out/r04e/b.err || { tail -3 gpurun_out/r04e/b.err; exit 1; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04e/b.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], "direct", d["roofline"]["by_kernel"]["direct"]["ms_per_step"], flush=True)
PY
done
