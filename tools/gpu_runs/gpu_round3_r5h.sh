# the driver's launch form (torch.distributed.run, one rank) with the RCCL choreography forced on: rendezvous from the
# environment, device binding, comm-launch stream, bucket collectives, per-bucket timeline
mkdir -p gpurun_out/r5h
DPAMD_DDP_FORCE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 4 > gpurun_out/r5h/torchrun_forced.log 2>&1; rc=$?
tail -1 gpurun_out/r5h/torchrun_forced.log | cut -c1-1800; exit $rc
