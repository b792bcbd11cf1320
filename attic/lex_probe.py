import os, sys, json, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench_wfst as BW, b2t_native as N, ngram_lm, lm_decoder
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, *_ = BW.make(U=1)
_, _, _, _, seqs, logits, lens, _ = BW.make(U=4, seed=0, noise=0.3, graph=(prons, words, arpa, g), truth="lm")
lex = ngram_lm.Lexicon(prons, 41); wlm = ngram_lm.SparseNGramLM.from_arpa(arpa, lex.words)
for pen in (math.log(90.0), 0.0):
    lg = torch.from_numpy(logits).to(dev); pri = torch.zeros_like(lg); lp = torch.empty_like(lg)
    N.check(lib.b2t_lm_prologue_f32(lg.data_ptr(), pri.data_ptr(), float(pen), lp.data_ptr(), lg.shape[0] * lg.shape[1], 41, None), "p")
    for alpha, eos in ((3.08, True), (1.0, True), (0.3, False), (0.0, False)):
        opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.325, 1.0, 0.0, 100)
        opts.first_beam_size, opts.second_beam_size = 10, 100
        opts.lm_alpha, opts.lm_beta, opts.lm_eos = alpha, 0.0, eos
        res = lm_decoder.DecodeResource("", "", "", "", ""); res.set_lexicon_lm(lex, wlm, sil=1)
        for u in range(2):
            dec = lm_decoder.BrainSpeechDecoder(res, opts, max_len=logits.shape[1] + 8)
            dec.Decode(lp[u, :lens[u]])
            hyp = dec.result()
            print(f"pen {pen:.1f} alpha {alpha} eos {eos} u{u}: truth {' '.join(seqs[u])} | hyp {hyp[0].sentence if hyp else None} | n {len(hyp)}")
    am = lp[0, :lens[0]].argmax(-1).cpu().numpy()
    print("greedy", am.tolist()); print("truth phones", [list(prons[w][0]) for w in seqs[0]])
