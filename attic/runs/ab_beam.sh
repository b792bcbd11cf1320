cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python -c "
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, 'nejm-brain-to-text_amd')
import bench_secondary as S
d = S.decode_beam100_3gram(); s = S.stream_32utt_5gram()
print('beam100 ms/utt', d['p50_ms_per_utterance'], 'call', d['ms_per_call_32_utterances'], '| stream p50', s['p50_ms_per_frame'])" 2>/dev/null | tail -1; }
for t in 256 512 1024; do echo "threads $t: $(run B2T_BEAM_THREADS=$t)"; done
echo "default: $(run A=1)"
