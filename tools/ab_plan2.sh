cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "default (80 TF, 5.5/6.0): $(run A=1)"
echo "gemm 50 TF:               $(run B2T_EST_GEMM_TFS=50)"
echo "gemm 35 TF:               $(run B2T_EST_GEMM_TFS=35)"
echo "gemm 120 TF:              $(run B2T_EST_GEMM_TFS=120)"
echo "bwd 7.2 us:               $(run B2T_EST_BWD_US=7.2)"
echo "gemm 50, bwd 7.2:         $(run B2T_EST_GEMM_TFS=50 B2T_EST_BWD_US=7.2)"
echo "gemm 35, bwd 7.2, fwd 6:  $(run B2T_EST_GEMM_TFS=35 B2T_EST_BWD_US=7.2 B2T_EST_FWD_US=6)"
echo "default:                  $(run A=1)"
