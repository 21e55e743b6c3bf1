cd $GRAFT_REPO_ROOT
for ms in 12 8 5 3 1; do
  echo "== T3D_WG_MIN_STEPS=$ms"
  for shp in "50176 196 192 64" "50176 196 384 64" "50176 196 384 96" "50176 196 576 96" "12544 49 576 160" "12544 49 960 160" "12544 49 960 320" "12544 49 160 960" "12544 49 320 1280" "50176 196 480 112" "12544 49 672 160" "200704 784 144 32" "200704 784 192 32"; do
    T3D_WG_MIN_STEPS=$ms python tools/run_kernel.py pwwgrad $shp --reps 20 --nrep 16 2>&1 | tail -1
  done
done
