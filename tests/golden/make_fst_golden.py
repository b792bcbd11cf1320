#!/usr/bin/env python3
"""Pins the token FST T and the lexicon FST L of the decode graph to the reference's OWN graph tools, run here on a toy
dictionary exactly as the recipe chains them (language_model/examples/speech/s0/run.sh:43-45):

  local/build_lm.sh:22-27            local/remove_stress_marker.py  dict -> lm/dict
  local/prepare_dict_ctc.sh:28-50    first pronunciation per word (the perl one-liner), sort | uniq -> lexicon.txt,
                                     units.txt from local/all_phoneme_units.txt
  tools/fst/ctc_compile_dict_token.sh:55-98
                                     lexiconp.txt (perl -ape), tools/fst/add_lex_disambig.pl, tokens.txt
                                     (<eps> <blk> SIL units #0..#N), tools/fst/ctc_token_fst_corrected.py decode tokens.txt
                                     (= T in text form), words.txt (awk), tools/fst/make_lexicon_fst.pl --pron-probs
                                     lexiconp_disambig.txt $sil_prob SIL '#'$ndisambig (= L in text form, before
                                     fstaddselfloops)

Nothing of the reference is copied: its scripts are EXECUTED where they lie under /root/reference (python / perl / awk /
sort are in the image; fstcompile / fstaddselfloops / fstarcsort are not, so the fixture holds the text forms the
scripts print, with the symbols mapped through tokens.txt / words.txt the way `fstcompile --isymbols --osymbols` would).
Writes tests/golden/fst_tl.npz (data only).  tests/test_fst_golden.py compares wfst.token_fst / wfst.lex_disambig /
wfst.lexicon_fst_disambig (and the native graph compiler's inputs) against it on any machine.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/language_model"
S0 = os.path.join(REF, "examples", "speech", "s0")
FST = os.path.join(REF, "tools", "fst")

# a CMUdict-shaped toy dictionary: stress markers, homophones (TWO / TOO / TO share T UW), a pronunciation that is a prefix
# of another (A = AH, AN = AH N, AND = AH N D), a second pronunciation of a word (dropped by the recipe), repeated phones,
# one-phone words, and an entry whose only phone sequence repeats another word's.
TOY_DICT = """A  AH0
A  EY1
AN  AH0 N
AND  AH0 N D
ANDY  AE1 N D IY0
TWO  T UW1
TOO  T UW1
TO  T UW1
TOOL  T UW1 L
TOOLS  T UW1 L Z
I  AY1
EYE  AY1
AYE  AY1
ICE  AY1 S
NICE  N AY1 S
NIECE  N IY1 S
KNEES  N IY1 Z
SEE  S IY1
SEA  S IY1
SEAT  S IY1 T
SEATTLE  S IY0 AE1 T AH0 L
MISSISSIPPI  M IH2 S IH0 S IH1 P IY0
THE  DH AH0
THE  DH IY0
THEE  DH IY1
ZOO  Z UW1
OH  OW1
OWE  OW1
"""


def run(cmd, cwd, stdin=None):
    r = subprocess.run(cmd, cwd=cwd, input=stdin, capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"))
    if r.returncode != 0:
        raise RuntimeError(f"{cmd}: {r.stderr[-800:]}")
    return r.stdout


def recipe(tmp, sil_prob):
    """-> dict of arrays for one sil_prob"""
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, "dict_in"), "w") as f:
        f.write(TOY_DICT)
    # build_lm.sh:22-27
    run([sys.executable, os.path.join(S0, "local", "remove_stress_marker.py"), "dict_in", "dict"], tmp)
    # prepare_dict_ctc.sh:28-31 (the one-liner, verbatim semantics: first pronunciation of a word wins), :47 sort | uniq
    raw = run(["perl", "-e", 'while(<>){@A = split; if(! $seen{$A[0]}) {$seen{$A[0]} = 1; $s = join(" ",@A); print $s; print "\\n"}}',
               "dict"], tmp)
    lexicon = run(["sort"], tmp, stdin=raw)
    lexicon = run(["uniq"], tmp, stdin=lexicon)
    with open(os.path.join(tmp, "lexicon.txt"), "w") as f:
        f.write(lexicon)
    units = open(os.path.join(S0, "local", "all_phoneme_units.txt")).read().split()
    # ctc_compile_dict_token.sh:55
    lexp = run(["perl", "-ape", r's/(\S+\s+)(.+)/${1}1.0\t$2/;'], tmp, stdin=lexicon)
    with open(os.path.join(tmp, "lexiconp.txt"), "w") as f:
        f.write(lexp)
    # :59-60 (called WITHOUT --pron-probs, as the recipe does: the "1.0" column counts as a phone of every entry)
    nd = int(run(["perl", os.path.join(FST, "add_lex_disambig.pl"), "lexiconp.txt", "lexiconp_disambig.txt"], tmp).strip())
    ndisambig = nd + 1
    # :62-67 tokens.txt
    tokens = ["<eps>", "<blk>", "SIL"] + units + [f"#{n}" for n in range(ndisambig + 1)]
    with open(os.path.join(tmp, "tokens.txt"), "w") as f:
        f.write("".join(f"{t} {i}\n" for i, t in enumerate(tokens)))
    # :74 T, text form (labels are already numbers)
    t_text = run([sys.executable, os.path.join(FST, "ctc_token_fst_corrected.py"), "decode", "tokens.txt"], tmp)
    # :77-87 words.txt
    words_txt = run(["awk", """
  BEGIN { print "<eps> 0"; }
  { printf("%s %d\\n", $1, NR); }
  END { printf("#0 %d\\n", NR+1); printf("<s> %d\\n", NR+2); printf("</s> %d\\n", NR+3); }"""], tmp,
                    stdin=run(["uniq"], tmp, stdin=run(["sort"], tmp, stdin=run(["awk", "{print $1}"], tmp, stdin=lexp))))
    words = [l.split()[0] for l in words_txt.splitlines()]
    assert [int(l.split()[1]) for l in words_txt.splitlines()] == list(range(len(words)))
    # :94 L, text form with symbols
    l_text = run(["perl", os.path.join(FST, "make_lexicon_fst.pl"), "--pron-probs", "lexiconp_disambig.txt", str(sil_prob), "SIL",
                  f"#{ndisambig}"], tmp)
    tok_id = {t: i for i, t in enumerate(tokens)}
    word_id = {w: i for i, w in enumerate(words)}

    def parse(text, isym, osym):
        arcs, finals = [], []
        for line in text.splitlines():
            p = line.split()
            if len(p) <= 2:
                finals.append((int(p[0]), float(p[1]) if len(p) == 2 else 0.0))
            else:
                il = isym[p[2]] if isym else int(p[2])
                ol = osym[p[3]] if osym else int(p[3])
                arcs.append((int(p[0]), int(p[1]), il, ol, float(p[4]) if len(p) > 4 else 0.0))
        a = np.array(arcs, dtype=np.float64).reshape(-1, 5)
        return a, np.array(finals, dtype=np.float64).reshape(-1, 2)

    t_arcs, t_final = parse(t_text, None, None)
    l_arcs, l_final = parse(l_text, tok_id, word_id)
    dis_lines = open(os.path.join(tmp, "lexiconp_disambig.txt")).read().splitlines()
    return dict(tokens=np.array(tokens), words=np.array(words), lexicon=np.array(lexicon.splitlines()),
                lexicon_disambig=np.array(dis_lines), ndisambig=np.int64(ndisambig),
                t_arcs=t_arcs, t_final=t_final, l_arcs=l_arcs, l_final=l_final, sil_prob=np.float64(sil_prob))


def main():
    out = {"toy_dict": np.array(TOY_DICT)}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, sp in (("p5", 0.5), ("p2", 0.2), ("p0", 0)):
            for k, v in recipe(os.path.join(tmp, tag), sp).items():
                out[f"{tag}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "fst_tl.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", ""), v.dtype)


if __name__ == "__main__":
    main()
