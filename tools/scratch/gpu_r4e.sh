set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "-- rocm runtime"; timeout -k 10 120 python tools/pipe_probe.py 20
echo "-- torch runtime"; FUZZYHIP_HIP_RUNTIME=torch timeout -k 10 120 python tools/pipe_probe.py 20
echo "-- torch runtime again"; FUZZYHIP_HIP_RUNTIME=torch timeout -k 10 120 python tools/pipe_probe.py 20
