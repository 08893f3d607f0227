set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6c26
NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rccl_two_ranks_one_gpu.py 2>&1 | grep -i "refused\|duplicate\|all_reduce over\|invalid usage" | cut -c1-700 | head -12 | tee gpurun_out/r6c26/probe.txt
