"""Back-off n-gram LM: the HBM automaton built by the host (ngram_lm.NGramLM) scores exactly like the oracle's
dictionary back-off (oracle.b2t_oracle.ngram_log10) on ARPA text; CPU only."""
import numpy as np
import pytest

from oracle import b2t_oracle as O

TOY = """\\data\\
ngram 1=5
ngram 2=4
ngram 3=2

\\1-grams:
-99 <s> -0.5
-0.7 a -0.3
-0.5 b -0.2
-0.9 c -0.1
-1.2 </s>

\\2-grams:
-0.2 <s> a -0.4
-0.4 a b -0.25
-0.6 b </s>
-0.8 b c -0.15

\\3-grams:
-0.1 <s> a b
-0.3 a b c

\\end\\
"""
WORDS = [None, "a", "b", "c"]          # class 0 = blank


def test_arpa_hand_values():
    order, tab = O.parse_arpa(TOY)
    assert order == 3
    assert O.ngram_log10(order, tab, ("<s>",), "a") == pytest.approx(-0.2)
    assert O.ngram_log10(order, tab, ("<s>", "a"), "b") == pytest.approx(-0.1)
    # (a b) exists as context but has no "a" continuation: bow(a b) + [(b) exists, no (b a)] bow(b) + p(a)
    assert O.ngram_log10(order, tab, ("a", "b"), "a") == pytest.approx(-0.25 - 0.2 - 0.7)
    # unseen context (c c): skipped without weight, then bow(c) + p(b)
    assert O.ngram_log10(order, tab, ("c", "c"), "b") == pytest.approx(-0.1 - 0.5)


def test_automaton_matches_dictionary_backoff():
    import ngram_lm
    order, tab = O.parse_arpa(TOY)
    lm = ngram_lm.NGramLM.from_arpa(TOY, WORDS)
    rng = np.random.RandomState(0)
    for _ in range(200):
        n = rng.randint(1, 8)
        ids = rng.randint(1, 4, size=n)
        words = [WORDS[i] for i in ids]
        hist, ref = ["<s>"], 0.0
        for w in words + ["</s>"]:
            ref += O.ngram_log10(order, tab, tuple(hist), w) * np.log(10.0)
            hist.append(w)
        assert lm.sentence_logp(ids, bos=True, eos=True) == pytest.approx(ref, rel=1e-5, abs=1e-5)


@pytest.mark.parametrize("order", [2, 3, 5])
def test_synthetic_arpa_roundtrip(order):
    import ngram_lm
    words = [None] + [f"p{i}" for i in range(1, 41)]
    text = ngram_lm.synthetic_arpa(words, order, 300, seed=order)
    o2, tab = O.parse_arpa(text)
    assert o2 == order
    lm = ngram_lm.NGramLM.from_arpa(text, words)
    assert lm.order == order and lm.V == 44
    rng = np.random.RandomState(1)
    for _ in range(100):
        ids = rng.randint(1, 41, size=rng.randint(1, 12))
        hist, ref = ["<s>"], 0.0
        for i in ids:
            ref += O.ngram_log10(order, tab, tuple(hist), words[i]) * np.log(10.0)
            hist.append(words[i])
        assert lm.sentence_logp(ids, bos=True) == pytest.approx(ref, rel=1e-5, abs=1e-4)


def test_oracle_fused_beam_prefers_lm_consistent_hypothesis():
    order, tab = O.parse_arpa(TOY)
    # two frames, acoustic evidence slightly favours "a c", the LM strongly favours "a b"
    logp = np.log(np.array([[0.1, 0.8, 0.05, 0.05], [0.1, 0.05, 0.40, 0.45]], dtype=np.float64)).astype(np.float32)
    plain = O.prefix_beam_search(logp, 4, 4)
    fused = O.prefix_beam_search_lm(logp, order, tab, WORDS, alpha=2.0, beta=0.0, first_beam=4, second_beam=4)
    assert plain[0][0] == (1, 3)
    assert fused[0][0] == (1, 2)


def test_word_level_oracle_and_host_replay_agree():
    """Lexicon-constrained oracle search vs the host replay of word emissions (ngram_lm.replay_words) on the oracle's
    own hypotheses: same words, same LM score."""
    import ngram_lm
    prons = ngram_lm.synthetic_lexicon(120, 12, seed=3)
    lex = ngram_lm.Lexicon(prons, 12)
    text = ngram_lm.synthetic_word_arpa(lex.words, 3, 300, seed=4)
    order, tab = O.parse_arpa(text)
    lm = ngram_lm.SparseNGramLM.from_arpa(text, lex.words)
    rng = np.random.default_rng(0)
    logp = O.log_softmax((rng.standard_normal((40, 12)) * 1.5).astype(np.float32))
    res = O.prefix_beam_search_lexicon(logp, prons, order, tab, 0.7, 0.4, 8, 12)
    assert res, "the oracle found no complete hypothesis"
    for p, words, ctc, lmscore, total in res:
        w2, l2 = ngram_lm.replay_words(lex, lm, p, 0.7, 0.4)
        assert tuple(w2) == words
        assert l2 == pytest.approx(lmscore, rel=1e-5, abs=1e-4)
