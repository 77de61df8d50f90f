# Round 5, VERDICT r4 item 3: the bounded go / no-go on the persistent EDSR block chain.  16 residual blocks, 32 x 48 x 48, training plan (T stored):
# per-block launches against the chain with (0) the memory-side hand-off, (5) the XCD-local hand-off, (1) NO hand-off at all (wrong results: the bound
# of ANY hand-off), (3) no hand-off and no T stores (the inference form's bound).      usage (GPU box): bash tests/tools/r05_chain.sh
cd $GRAFT_REPO_ROOT
for v in 0 5 1 3; do
  echo "=== CHAIN_ABL=$v"
  if [ $v = 0 ]; then python3 tests/tools/chain_ab.py 30 2>&1 | grep -v amdgpu.ids; else RUMPY_EXP_LIB=$GRAFT_REPO_ROOT/build_abl/CHAIN_$v/librumpy_exp.so python3 tests/tools/chain_ab.py 30 2>&1 | grep -v amdgpu.ids; fi
done
python3 tests/tools/kbench.py block 2>&1 | grep -v amdgpu.ids | tail -6
