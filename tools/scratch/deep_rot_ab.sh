cd $GRAFT_REPO_ROOT
for e in "" "T3D_DEEP_NOROT=1"; do
  echo "== ${e:-rotated}"
  for args in "pwfwd 12544 49 960 160" "pwfwd 12544 49 576 160" "pwfwd 12544 49 960 320" "pwdgrad 12544 49 160 960" "pwdgrad 12544 49 320 1280" "pwfwd 3136 49 4608 512" "pwfwd 3136 49 2048 512" "pwdgrad 3136 49 1024 2048"; do
    env $e python tools/run_kernel.py $args --reps 20 --nrep 16 --frag 2>&1 | tail -1
  done
done
