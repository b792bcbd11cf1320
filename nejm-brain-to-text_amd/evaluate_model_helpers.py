"""Evaluation helpers with the names evaluate_model.py imports from the reference module of the same
name (model_training/evaluate_model_helpers.py): the per-trial decoding step runs on the HIP path;
the phoneme table / logit re-ordering / text clean-up are format facts the decoder stage needs.

Not provided (I/O glue, SURVEY §2 row 5 "out of scope"): the Redis stream client functions
(reset/send/finalize_remote_language_model, update_remote_lm_params, get_current_redis_time_ms).
"""
import re

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
    with torch.no_grad():
        logits, _ = model(x=x, day_idx=torch.tensor([input_layer], device=device), states=None, return_state=True)
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
