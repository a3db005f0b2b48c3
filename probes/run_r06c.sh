#!/bin/bash
# round 4, closing call C: PMC passes (FETCH_SIZE, WRITE_SIZE) of the serialized base step on the closing tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash probes/run_pmc.sh r06a 2>&1 | tail -30
