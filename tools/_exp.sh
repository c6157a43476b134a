for v in H0 H1 M3H0 M2H3; do echo $v; PFNL_HIP_LIB=pfnl_amd/lib/var_$v.so timeout 200 tools/bench_brief.sh --conv3x3 winograd_ws; done
