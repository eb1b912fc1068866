cd $GRAFT_REPO_ROOT
UR_CHAIN_STAMPS=1 UR_LIB=$PWD/unirestore_amd/ab/lib_abl6.so python tools/bench_chain.py 2>&1 | tail -6
