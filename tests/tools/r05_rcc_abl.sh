cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 4 8 15; do for c in "fwd 32 48 48 20 4" "bwd 32 48 48 20 4"; do echo -n "abl $abl: "; RUMPY_RCC_ABL=$abl timeout 120 python3 tests/tools/rcab_chain_dbg.py $c 2>&1 | grep "chain:"; done; done
python3 tests/tools/kbench.py rcab 2>&1 | grep "us per"
