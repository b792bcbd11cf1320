cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "3 sweep streams, GEMMs on main, 6/4:      $(run B2T_STREAM_MAP='0,1,2,0,1;-1,-1,-1,-1,-1')"
M='-1,0,1,2,-1;-1,0,1,2,-1'
echo "cells (gemm+sweep of a layer on one queue; layers 0,4 on main) 6/4:   $(run B2T_STREAM_MAP=$M)"
echo "cells 8/4:    $(run B2T_STREAM_MAP=$M B2T_CHUNKS=8 B2T_CHUNKS_BWD=4)"
echo "cells 10/6:   $(run B2T_STREAM_MAP=$M B2T_CHUNKS=10 B2T_CHUNKS_BWD=6)"
echo "cells 12/8:   $(run B2T_STREAM_MAP=$M B2T_CHUNKS=12 B2T_CHUNKS_BWD=8)"
echo "cells 16/10:  $(run B2T_STREAM_MAP=$M B2T_CHUNKS=16 B2T_CHUNKS_BWD=10)"
M='0,1,2,-1,0;0,1,2,-1,0'
echo "cells (layers 0,4 on worker 0, layer 3 on main) 6/4:   $(run B2T_STREAM_MAP=$M)"
echo "cells' 10/6:   $(run B2T_STREAM_MAP=$M B2T_CHUNKS=10 B2T_CHUNKS_BWD=6)"
M='0,1,2,0,1;-1,-1,-1,-1,-1'
echo "3 sweep streams, GEMMs on main, 8/4:      $(run B2T_STREAM_MAP=$M B2T_CHUNKS=8)"
echo "3 sweep streams, GEMMs on main, 6/3:      $(run B2T_STREAM_MAP=$M B2T_CHUNKS_BWD=3)"
echo "3 sweep streams, GEMMs on main, 6/5:      $(run B2T_STREAM_MAP=$M B2T_CHUNKS_BWD=5)"
echo "3 sweep streams, GEMMs on main, 5/4:      $(run B2T_STREAM_MAP=$M B2T_CHUNKS=5)"
echo "3 sweep (narrow fwd), GEMMs on main, 6/4: $(run B2T_STREAM_MAP=$M B2T_WIDE_F32=)"
