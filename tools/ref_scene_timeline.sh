# Kernel timeline of one reference-scene BVH step (profiles/r02_bvh_build_timeline.txt): run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/ref_scene_steps.py 10 5 > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | sort | tail -1)
python tools/trace_timeline.py $f 9 > gpurun_out/timeline.txt
rm -rf gpurun_out/tl
tail -1 gpurun_out/timeline.txt
