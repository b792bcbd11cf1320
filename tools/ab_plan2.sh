cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for r in 1 2; do
echo "default:        $(run A=1)"
echo "dX splitk 2:    $(run B2T_DX_SPLITK=2)"
echo "dX splitk 3:    $(run B2T_DX_SPLITK=3)"
done
