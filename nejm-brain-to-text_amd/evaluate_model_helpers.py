"""Evaluation helpers with the names evaluate_model.py imports from the reference module of the same
name (model_training/evaluate_model_helpers.py): the per-trial decoding step runs on the HIP path;
the phoneme table / logit re-ordering / text clean-up are format facts the decoder stage needs.

The Redis stream client functions of the reference (reset/send/finalize_remote_language_model,
update_remote_lm_params, get_current_redis_time_ms: evaluate_model_helpers.py:129-296) speak the same streams and
field names here; `r` is a redis-py connection to a running LM server or a `remote_lm.LocalLMService` (the HIP beam
search in-process).
"""
import re
import time

import numpy as np
import torch

import b2t_ops as ops
from data_augmentations import gauss_smooth

# RNN output order: blank, 39 ARPAbet phonemes, silence (evaluate_model_helpers.py:9-20)
LOGIT_TO_PHONEME = ['BLANK'] + ('AA AE AH AO AW AY B CH D DH EH ER EY F G HH IH IY JH K L M N NG OW OY P R S SH T TH '
                                'UH UW V W Y Z ZH').split() + [' | ']


def rearrange_speech_logits_pt(logits):
    """[BLANK, phonemes..., SIL] -> [BLANK, SIL, phonemes...]: the LM decoder's token order
    (evaluate_model_helpers.py:79-83)."""
    return np.concatenate((logits[:, :, 0:1], logits[:, :, -1:], logits[:, :, 1:-1]), axis=-1)


def runSingleDecodingStep(x, input_layer, model, model_args, device):
    """Smooth one trial ('valid' padding) and run it through the model (evaluate_model_helpers.py:87-115).
    x [1,T,F] (any float dtype; the reference passes bf16) -> logits [1,T'',C] float32 numpy."""
    tr = model_args['dataset']['data_transforms']
    x = x.to(device)
    x = gauss_smooth(inputs=x, device=device, smooth_kernel_std=tr['smooth_kernel_std'],
                     smooth_kernel_size=tr['smooth_kernel_size'], padding='valid')
    import b2t_ops as ops
    was = ops.AMP["on"]
    ops.set_amp(ops.precision_from_args(model_args))   # autocast(enabled=model_args['use_amp']) at evaluate_model_helpers.py:90
    try:
        with torch.no_grad():
            logits, _ = model(x=x, day_idx=torch.tensor([input_layer], device=device), states=None, return_state=True)
    finally:
        ops.set_amp(was)
    return logits.float().cpu().numpy()


def greedy_phonemes(logits_t):
    """evaluate_model.py:129-141 display rule: argmax, drop blanks, then merge repeats."""
    a = np.argmax(logits_t, axis=-1)
    a = a[a != 0]
    if a.size:
        a = a[np.concatenate([[True], a[1:] != a[:-1]])]
    return [LOGIT_TO_PHONEME[int(p)] for p in a]


def remove_punctuation(sentence):
    s = re.sub(r"[^a-zA-Z\- ']", '', sentence)
    s = s.replace('- ', ' ').lower().replace('--', '').replace(" '", "'")
    return ' '.join(w for w in s.strip().split() if w)


def _extract_transcription(arr):
    end = int(np.argwhere(arr == 0)[0, 0])
    return ''.join(chr(int(c)) for c in arr[:end])


def load_h5py_file(file_path, b2txt_csv_df):
    """Read one session file into the dict-of-lists evaluate_model.py expects (evaluate_model_helpers.py:29-77)."""
    import h5py
    keys = ('neural_features', 'n_time_steps', 'seq_class_ids', 'seq_len', 'transcriptions', 'sentence_label',
            'session', 'block_num', 'trial_num', 'corpus')
    data = {k: [] for k in keys}
    with h5py.File(file_path, 'r') as f:
        for key in list(f.keys()):
            g = f[key]
            session, block_num = g.attrs['session'], g.attrs['block_num']
            y, m, d = session.split('.')[1:]
            row = b2txt_csv_df[(b2txt_csv_df['Date'] == f'{y}-{m}-{d}') & (b2txt_csv_df['Block number'] == block_num)]
            data['neural_features'].append(g['input_features'][:])
            data['n_time_steps'].append(g.attrs['n_time_steps'])
            data['seq_class_ids'].append(g['seq_class_ids'][:] if 'seq_class_ids' in g else None)
            data['seq_len'].append(g.attrs['seq_len'] if 'seq_len' in g.attrs else None)
            data['transcriptions'].append(g['transcription'][:] if 'transcription' in g else None)
            data['sentence_label'].append(g.attrs['sentence_label'][:] if 'sentence_label' in g.attrs else None)
            data['session'].append(session)
            data['block_num'].append(block_num)
            data['trial_num'].append(g.attrs['trial_num'])
            data['corpus'].append(row['Corpus'].values[0])
    return data


# ---- language model over Redis streams (client side) -----------------------------------------------------------
def get_current_redis_time_ms(redis_conn):
    sec, usec = redis_conn.time()
    return int(sec * 1000 + usec / 1000)


def _post_and_await(r, post_stream, fields, reply_stream, last_seen, what, pause=0.001):
    """xadd the request, then block on the reply stream until an entry newer than `last_seen` arrives.
    Returns (id of that entry, its fields)."""
    r.xadd(post_stream, fields)
    if pause:
        time.sleep(pause)
    while True:
        got = r.xread({reply_stream: last_seen}, count=1, block=10000)
        if got:
            entry_id, data = got[0][1][-1]
            return entry_id, data
        print(f'Still waiting for remote lm {what} from ts {last_seen}...')


def reset_remote_language_model(r, remote_lm_done_resetting_lastEntrySeen):
    seen, _ = _post_and_await(r, 'remote_lm_reset', {'done': 0}, 'remote_lm_done_resetting',
                              remote_lm_done_resetting_lastEntrySeen, 'reset')
    return seen


def update_remote_lm_params(r, remote_lm_done_updating_lastEntrySeen, acoustic_scale=0.35, blank_penalty=90.0,
                            alpha=0.55):
    seen, _ = _post_and_await(r, 'remote_lm_update_params',
                              {'acoustic_scale': acoustic_scale, 'blank_penalty': blank_penalty, 'alpha': alpha},
                              'remote_lm_done_updating_params', remote_lm_done_updating_lastEntrySeen,
                              'to update parameters')
    return seen


def send_logits_to_remote_lm(r, remote_lm_input_stream, remote_lm_output_partial_stream,
                             remote_lm_output_partial_lastEntrySeen, logits):
    seen, data = _post_and_await(r, remote_lm_input_stream, {'logits': np.float32(logits).tobytes()},
                                 remote_lm_output_partial_stream, remote_lm_output_partial_lastEntrySeen,
                                 'partial output', pause=0)
    return seen, data[b'lm_response_partial'].decode()


def finalize_remote_lm(r, remote_lm_output_final_stream, remote_lm_output_final_lastEntrySeen):
    """Returns (last entry id, dict of candidate lists sorted by total score, duplicates removed)."""
    seen, data = _post_and_await(r, 'remote_lm_finalize', {'done': 0}, remote_lm_output_final_stream,
                                 remote_lm_output_final_lastEntrySeen, 'final output', pause=0.005)
    parts = data[b'scoring'].decode().split(';') if data.get(b'scoring') else []
    cands = [(parts[i], float(parts[i + 1]), float(parts[i + 2]), float(parts[i + 3]), float(parts[i + 4]))
             for i in range(0, len(parts) - 4, 5)]
    if not cands:
        print('No candidate sentences were received from the language model.')
        cands = [('', 0, 0, 0, 0)]
    else:
        cands.sort(key=lambda c: c[4], reverse=True)       # total score, best first
        first = {}
        for c in cands:                                     # keep the best-scoring copy of a repeated sentence
            first.setdefault(c[0], c)
        cands = list(first.values())
    keys = ('candidate_sentences', 'candidate_acoustic_scores', 'candidate_ngram_scores', 'candidate_llm_scores',
            'candidate_total_scores')
    return seen, {k: [c[i] for c in cands] for i, k in enumerate(keys)}
