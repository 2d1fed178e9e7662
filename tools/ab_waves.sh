set -e
cd $GRAFT_REPO_ROOT/taichi_gaussian_rasterizer_amd
cp libgsplat_hip.so /tmp/base.so
for v in base w7 w8 base w7 w8; do
  if [ $v = base ]; then cp /tmp/base.so libgsplat_hip.so; else cp libgsplat_hip_$v.so libgsplat_hip.so; fi
  cd ..; echo -n "$v "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms']['gs_raster_bwd'])"; cd taichi_gaussian_rasterizer_amd
done
cp /tmp/base.so libgsplat_hip.so
