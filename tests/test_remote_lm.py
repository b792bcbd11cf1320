"""The Redis-stream exchange between evaluate_model.py and the language model (reference:
model_training/evaluate_model_helpers.py:129-296, language_model/language-model-standalone.py:520-790), served
in-process by remote_lm.LocalLMService.  CPU: protocol + wire format with a scripted decoder; GPU: the HIP beam search."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))


class _Res:
    def __init__(self, s, a, l):
        self.sentence, self.ac_score, self.lm_score = s, a, l


class _ScriptedDecoder:
    def __init__(self):
        self.n_reset, self.frames, self.finished = 0, 0, 0
        self._res = []

    def Reset(self):
        self.n_reset += 1; self.frames = 0; self._res = []

    def FinishDecoding(self):
        self.finished += 1

    def result(self):
        return self._res


def test_stream_protocol_and_wire_format():
    import evaluate_model_helpers as H
    from remote_lm import LocalLMService
    dec = _ScriptedDecoder()
    seen_calls = []

    def decode_fn(d, logits, priors, log_bp):
        seen_calls.append((logits.shape, float(log_bp)))
        d.frames += logits.shape[0]
        d._res = [_Res("the cat", -10.0, -3.0), _Res("the cat", -12.0, -3.5), _Res("a cat", -9.0, -6.0), _Res("the hat", -30.0, -2.0)]

    r = LocalLMService(dec, acoustic_scale=0.5, blank_penalty=90.0, decode_fn=decode_fn)
    r.flushall()
    t0 = H.get_current_redis_time_ms(r)
    partial_seen = final_seen = reset_seen = upd_seen = t0
    for trial in range(2):     # ids keep growing across trials, as evaluate_model.py relies on
        reset_seen = H.reset_remote_language_model(r, reset_seen)
        assert dec.n_reset == trial + 1
        upd_seen = H.update_remote_lm_params(r, upd_seen, acoustic_scale=0.5, blank_penalty=45.0, alpha=0.55)
        logits = np.random.default_rng(trial).standard_normal((17, 41)).astype(np.float32)
        partial_seen, decoded = H.send_logits_to_remote_lm(r, 'remote_lm_input', 'remote_lm_output_partial', partial_seen, logits)
        assert decoded == "the cat" and dec.frames == 17
        assert seen_calls[-1][0] == (17, 41) and abs(seen_calls[-1][1] - np.log(45.0)) < 1e-6
        final_seen, out = H.finalize_remote_lm(r, 'remote_lm_output_final', final_seen)
        assert dec.finished == trial + 1
        # sorted by total = 0.5 * ac + lm, duplicates collapsed onto their best copy
        assert out['candidate_sentences'] == ["the cat", "a cat", "the hat"]
        np.testing.assert_allclose(out['candidate_total_scores'], [-8.0, -10.5, -17.0])
        np.testing.assert_allclose(out['candidate_acoustic_scores'], [-10.0, -9.0, -30.0])
        np.testing.assert_allclose(out['candidate_ngram_scores'], [-3.0, -6.0, -2.0])
        assert out['candidate_llm_scores'] == [0.0, 0.0, 0.0]
    # the raw reply is what the reference's parser expects: 5 ';'-separated fields per candidate, bytes keys
    eid, fields = r.streams['remote_lm_output_final'][-1]
    assert len(fields[b'scoring'].decode().split(';')) == 5 * 4 and fields[b'lm_response_final'] == b"the cat"
    assert eid > t0


def test_no_candidates_edge_case():
    import evaluate_model_helpers as H
    from remote_lm import LocalLMService
    r = LocalLMService(_ScriptedDecoder(), decode_fn=lambda *a: None)
    seen, out = H.finalize_remote_lm(r, 'remote_lm_output_final', H.get_current_redis_time_ms(r))
    assert out['candidate_sentences'] == [''] and out['candidate_total_scores'] == [0]


@pytest.mark.gpu
def test_local_service_with_hip_decoder():
    """evaluate_model.py's per-trial sequence against the HIP prefix beam search + lexicon / word n-gram."""
    import evaluate_model_helpers as H
    import lm_decoder, ngram_lm
    from remote_lm import LocalLMService
    Cc = 41
    prons = ngram_lm.synthetic_lexicon(200, Cc, seed=5)
    lex = ngram_lm.Lexicon(prons, Cc)
    wlm = ngram_lm.SparseNGramLM.from_arpa(ngram_lm.synthetic_word_arpa(lex.words, 2, 400, seed=2), lex.words)
    res = lm_decoder.DecodeResource("", "", "", "", "")
    res.set_lexicon_lm(lex, wlm, sil=1)
    opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.35, 0.95, 0.0, 10)
    opts.lm_alpha, opts.lm_beta = 0.8, 0.0
    dec = lm_decoder.BrainSpeechDecoder(res, opts, max_len=128)
    r = LocalLMService(dec, acoustic_scale=0.35, blank_penalty=9.0, nbest=10)
    rs = np.random.RandomState(0)
    words = [lex.words[i] for i in rs.randint(0, 200, size=4)]
    frames = []
    for w in words:
        for c in list(prons[w][0]) + [1]:
            frames += [c, 0]
    lg = np.full((len(frames), Cc), -4.0, dtype=np.float32)
    for t, c in enumerate(frames):
        lg[t, c] = 6.0
    seen = H.get_current_redis_time_ms(r)
    s1 = H.reset_remote_language_model(r, seen)
    s2, partial = H.send_logits_to_remote_lm(r, 'remote_lm_input', 'remote_lm_output_partial', seen, lg)
    s3, out = H.finalize_remote_lm(r, 'remote_lm_output_final', seen)
    assert out['candidate_sentences'][0] == " ".join(words).lower() == partial
    tot = out['candidate_total_scores']
    assert all(tot[i] >= tot[i + 1] for i in range(len(tot) - 1))
    assert len(set(out['candidate_sentences'])) == len(out['candidate_sentences'])
