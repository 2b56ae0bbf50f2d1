set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 --share-gpus --size 4096 > $O/multi2.json 2> $O/multi2.err; echo rc=$?; tail -1 $O/multi2.json | cut -c1-400
python bench.py --gpus 2 --steps 3 --warmup 1 --share-gpus --config 4 --images 8 > $O/multi2_c4.json 2> $O/multi2_c4.err; echo rc=$?; tail -1 $O/multi2_c4.json | cut -c1-300
python bench.py --gpus 2 --steps 3 --warmup 1 --share-gpus --config 5 --size 4096 > $O/multi2_c5.json 2> $O/multi2_c5.err; echo rc=$?; tail -1 $O/multi2_c5.json | cut -c1-300
python bench.py --gpus 2 --steps 3 --warmup 1 > $O/multi2_refuse.json 2> $O/multi2_refuse.err; echo "rc=$? (expected 2: refuses fewer GPUs than asked)"; tail -1 $O/multi2_refuse.err | cut -c1-200
