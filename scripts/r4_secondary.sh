# BASELINE configs 3 - 5 + the other torchvision backbones on the sources in the tree (builder-run, not driver-timed)
O=gpurun_out/r4_secondary
mkdir -p $O
for w in densenet169-metablock davit-tiny-gfcam beitv2-large-bert-rgatt vgg16-crossattention mobilenetv2-crossattention efficientnetb0-crossattention; do
  timeout -k 10 400 python3 bench.py --workload $w --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err || { tail -3 $O/bench_$w.err; continue; }
  python -c "import sys,json; d=json.loads(open('$O/bench_$w.json').readlines()[-1]); print('$w', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
