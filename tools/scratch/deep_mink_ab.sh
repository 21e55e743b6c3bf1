cd $GRAFT_REPO_ROOT
for e in "T3D_X=0" "T3D_DEEP_MINK=128"; do
  echo "== $e"
  for args in "pwfwd 50176 196 384 64" "pwfwd 50176 196 384 96" "pwfwd 50176 196 576 96" "pwfwd 50176 196 192 64" "pwdgrad 50176 196 64 384" "pwdgrad 50176 196 96 576" "pwfwd 200704 784 192 32" "pwfwd 200704 784 144 32" "pwdgrad 200704 784 32 192" "pwfwd 802816 3136 144 24"; do
    env $e python tools/run_kernel.py $args --reps 20 --nrep 16 --frag 2>&1 | tail -1
  done
done
