# Functional rehearsal of the N > 1 path on a 1-GPU box: two ranks share the GPU, collectives over gloo (RCCL refuses two ranks
# on one device).  Exercises bench.py's multi-rank control flow end to end with the real kernels: replicas + fence + max over
# ranks for the forward metric, GradAllReducer (bucketed, hook-launched all-reduce) + scalar loss averaging for both training
# steps.  Not a measurement.   usage (through gpurun): bash tools/rehearse_two_ranks.sh
cd ${GRAFT_REPO_ROOT:-.}
SEGMIF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --train-batch 2 --train-steps 2 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*" | tail -3
