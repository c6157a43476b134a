for v in P00 P20 P02 P11; do echo $v; PFNL_HIP_LIB=pfnl_amd/lib/var_$v.so timeout 200 tools/bench_brief.sh --conv3x3 winograd_ws; done
