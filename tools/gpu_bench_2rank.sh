# Rehearsal of the 2-rank bench path on a 1-GPU box: both ranks share GPU 0, gloo for the gather.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export CUTDET_BENCH_BACKEND=gloo CUTDET_BENCH_ONE_GPU=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 4 --clip-frames 5000 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo rc=$?; tail -5 gpurun_out/bench2.err; cat gpurun_out/bench2.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 40 --warmup 4 --clip-frames 5000 --independent-clips > gpurun_out/bench2i.json 2> gpurun_out/bench2i.err; echo rc=$?; tail -3 gpurun_out/bench2i.err; cat gpurun_out/bench2i.json
