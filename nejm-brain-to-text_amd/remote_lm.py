"""In-process stand-in for the pair "Redis server + language-model-standalone.py" of the reference.

evaluate_model.py (model_training/evaluate_model.py:159-235) talks to the language model through Redis streams:
it posts to `remote_lm_reset` / `remote_lm_input` / `remote_lm_finalize` / `remote_lm_update_params` and reads the
replies from `remote_lm_done_resetting` / `remote_lm_output_partial` / `remote_lm_output_final` /
`remote_lm_done_updating_params`; the serving loop is language_model/language-model-standalone.py:520-790.
`LocalLMService` offers the subset of the redis-py client that this exchange uses (`time`, `flushall`, `xadd`, `xread`,
`get`) and answers every request synchronously, inside `xadd`, with an `lm_decoder.BrainSpeechDecoder` (the HIP prefix
beam search): `r = LocalLMService(decoder)` in place of `redis.Redis(...)` and the rest of evaluate_model.py runs
unchanged, with the wire format of the reference -- stream entries are `(id, {bytes: bytes})`, the final reply carries
`scoring` = `sentence;acoustic;ngram;llm;total` per candidate joined by `;` (language-model-standalone.py:650-659).
Not reproduced: n-best augmentation and OPT rescoring (llm score is 0.0, as in the reference's do_opt = 0 branch).
"""
import time
from typing import Callable, Dict, List, Optional

import numpy as np

INPUT_STREAM = "remote_lm_input"
PARTIAL_STREAM = "remote_lm_output_partial"
FINAL_STREAM = "remote_lm_output_final"


def _b(v) -> bytes:
    if isinstance(v, bytes):
        return v
    if isinstance(v, (bytearray, memoryview)):
        return bytes(v)
    return str(v).encode()


class LocalLMService:
    """decoder: object with Reset() / FinishDecoding() / result() (entries with .sentence, .ac_score, .lm_score).
    decode_fn(decoder, logits[T, C], log_priors, log_blank_penalty): defaults to lm_decoder.DecodeNumpy (HIP path)."""

    def __init__(self, decoder, n_classes: int = 41, acoustic_scale: float = 0.35, blank_penalty: float = 90.0,
                 alpha: float = 0.55, nbest: int = 100, input_stream: str = INPUT_STREAM,
                 partial_output_stream: str = PARTIAL_STREAM, final_output_stream: str = FINAL_STREAM,
                 decode_fn: Optional[Callable] = None):
        self.decoder = decoder
        self.n_classes = n_classes
        self.params = dict(acoustic_scale=float(acoustic_scale), blank_penalty=float(blank_penalty), alpha=float(alpha),
                           nbest=int(nbest))
        self.input_stream, self.partial_stream, self.final_stream = input_stream, partial_output_stream, final_output_stream
        if decode_fn is None:
            import lm_decoder
            decode_fn = lm_decoder.DecodeNumpy
        self.decode_fn = decode_fn
        self.streams: Dict[str, List] = {}
        self.kv: Dict[str, bytes] = {}
        self._last_id = 0

    # ---- the redis-py calls the exchange uses ------------------------------------------------------------------
    def time(self):
        t = time.time()
        return int(t), int((t - int(t)) * 1e6)

    def flushall(self):
        self.streams.clear()
        self.kv.clear()

    def get(self, key):
        return self.kv.get(key)

    def set(self, key, value):
        self.kv[key] = _b(value)

    def _append(self, stream: str, fields: dict) -> int:
        sec, usec = self.time()
        self._last_id = max(self._last_id + 1, sec * 1000 + usec // 1000)   # ids grow strictly; comparable with ms times
        self.streams.setdefault(stream, []).append((self._last_id, {_b(k): _b(v) for k, v in fields.items()}))
        return self._last_id

    def xadd(self, stream, fields):
        eid = self._append(stream, fields)
        self._serve(stream, self.streams[stream][-1][1])
        return eid

    def xread(self, streams: dict, count=None, block=None):
        out = []
        for name, last in streams.items():
            last = int(last.decode().split("-")[0]) if isinstance(last, bytes) else int(last)
            new = [e for e in self.streams.get(name, []) if e[0] > last]
            if new:
                out.append([_b(name), new[:count] if count else new])
        return out

    # ---- the serving loop of language-model-standalone.py, one request at a time ---------------------------------
    def _serve(self, stream: str, fields: dict):
        if stream == "remote_lm_reset":
            self.decoder.Reset()
            self._append("remote_lm_done_resetting", {"done": 1})
        elif stream == "remote_lm_update_params":
            for k in ("acoustic_scale", "blank_penalty", "alpha"):
                if _b(k) in fields:
                    self.params[k] = float(fields[_b(k)])
            if b"nbest" in fields:
                self.params["nbest"] = int(fields[b"nbest"])
            self._append("remote_lm_done_updating_params", {"done": 1})
        elif stream == self.input_stream:
            logits = np.frombuffer(fields[b"logits"], dtype=np.float32).reshape(-1, self.n_classes).copy()
            self.decode_fn(self.decoder, logits, np.zeros_like(logits), float(np.log(self.params["blank_penalty"])))
            res = self.decoder.result()
            self._append(self.partial_stream, {"lm_response_partial": res[0].sentence if res else ""})
        elif stream == "remote_lm_finalize":
            self.decoder.FinishDecoding()
            res = self.decoder.result()[: max(1, self.params["nbest"])]
            a = self.params["acoustic_scale"]
            scoring = ";".join(";".join(map(str, [d.sentence.strip(), d.ac_score, d.lm_score, 0.0, a * d.ac_score + d.lm_score]))
                               for d in res)
            reply = {"lm_response_final": res[0].sentence if res else ""}
            if self.params["nbest"] > 1:
                reply.update(scoring=scoring, context_str="")
            self._append(self.final_stream, reply)
            self._append("remote_lm_done_finalizing", {"done": 1})
