cd $GRAFT_REPO_ROOT
for cfg in "0 0" "100000 0" "0 512" "100000 512" "0 384" "100000 384"; do
  set -- $cfg
  echo "== T3D_YF_SK_M=$1 T3D_YF_S=$2"
  for shp in "3211264 12544 16 96" "802816 3136 24 144" "200704 784 32 192"; do
    T3D_YF_SK_M=$1 T3D_YF_S=$2 python tools/run_kernel.py pwbwd_yf $shp --reps 20 --nrep 16 2>&1 | tail -1
  done
done
