#!/bin/bash
# closing tree: serialized kernel traces of the large and video configurations (base: r04a)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
bash probes/run_prof.sh r04b large video > gpurun_out/r04b/prof.log 2>&1
head -14 gpurun_out/r04b/large_serialized_kernel_stats.txt | cut -c1-140; head -8 gpurun_out/r04b/video_serialized_kernel_stats.txt | cut -c1-140
