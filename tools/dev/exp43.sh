cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/avail.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/avail.txt
