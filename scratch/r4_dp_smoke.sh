# bench.py's data-parallel path end to end, several ranks on ONE GPU over gloo (HUGS_FORCE_DEVICE / HUGS_DIST_BACKEND test hooks)
export HUGS_FORCE_DEVICE=0 HUGS_DIST_BACKEND=gloo
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 5 --warmup 3 --min-time 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^$" | cut -c1-420 | tail -3
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 5 --warmup 3 --min-time 0 --scaling strong 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^$" | cut -c1-420 | tail -3
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29710 bench.py --gpus 8 --steps 5 --warmup 3 --min-time 0 --scaling strong 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^$" | cut -c1-420 | tail -3
