"""CPU oracle for the recurrent (LSTM sequence-to-sequence) decoder forward pass.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of a benchmark may import this module, and only as the
checker / the reported CPU baseline.  The product package never imports it.

Parity status: **parity unpinned**
----------------------------------
``src/rnn_viterbi.py`` cannot be imported (``Optional`` is used at :35 and never
imported), it feeds 3-D arrays (``prepare_batch`` :361-392) into
2-D placeholders (:263-279), and its arithmetic lives in TensorFlow 1.x
(``tf.contrib.rnn.LSTMCell`` :289,302, ``tf.nn.dynamic_rnn`` :292,305,
``tf.contrib.layers.linear`` :315), which is neither vendored nor installed
here; the reference's tests hold no vector for it.  This file restates the graph
AS WRITTEN in ``build_seq2seq_model`` (:235-340) with the published semantics of
those TF-1.x ops (tensorflow/python/ops/rnn_cell_impl.py, ``LSTMCell.call``, r1.15):

    lstm_matrix = concat([inputs, m_prev], 1) @ kernel + bias     kernel [1 + H, 4H]
    i, j, f, o  = split(lstm_matrix, 4, axis=1)
    c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j)   forget_bias = 1.0
    m = sigmoid(o) * tanh(c)

no peepholes, no projection, no clipping (the defaults the reference uses).
``dynamic_rnn(time_major=True)`` runs the cell over axis 0 from a zero state
(encoder) or from the encoder's final (c, m) (decoder, ``initial_state=`` :308);
``linear`` is ``outputs @ w + b`` on the last axis; ``tf.argmax(logits, 2)`` over the
size-1 vocabulary axis (:321, ``vocab_size = 1`` :71) is identically 0.
"""
from __future__ import annotations

import numpy as np


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def lstm_cell(x_t, c, m, kernel, bias, forget_bias=1.0):
    """One ``LSTMCell.call``: x_t [B, n_in], c/m [B, H], kernel [n_in + H, 4H], bias [4H]."""
    H = m.shape[1]
    z = np.matmul(np.concatenate([x_t, m], axis=1), kernel) + bias
    i, j, f, o = z[:, 0:H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:4 * H]
    c_new = _sigmoid(f + np.asarray(forget_bias, z.dtype)) * c + _sigmoid(i) * np.tanh(j)
    m_new = _sigmoid(o) * np.tanh(c_new)
    return c_new, m_new


def dynamic_rnn(inputs_tm, kernel, bias, state=None, forget_bias=1.0):
    """``tf.nn.dynamic_rnn(cell, inputs[T, B, n_in], time_major=True)`` -> (outputs [T, B, H], (c, m))."""
    T, B = inputs_tm.shape[0], inputs_tm.shape[1]
    H = kernel.shape[1] // 4
    dt = inputs_tm.dtype
    c, m = (np.zeros((B, H), dt), np.zeros((B, H), dt)) if state is None else state
    out = np.empty((T, B, H), dt)
    for t in range(T):
        c, m = lstm_cell(inputs_tm[t], c, m, kernel, bias, forget_bias)
        out[t] = m
    return out, (c, m)


def seq2seq_forward(encoder_inputs, decoder_inputs, params, dtype=np.float32):
    """build_seq2seq_model's inference tensors (src/rnn_viterbi.py:282-321).

    ``encoder_inputs`` [T_enc, B], ``decoder_inputs`` [T_dec, B] (time-major, as the
    placeholders are declared :263-279); ``params``: enc_kernel [1+H, 4H], enc_bias [4H],
    dec_kernel, dec_bias, proj_w [H, 1] or [H], proj_b [1].
    Returns ``(decoder_logits [T_dec, B], decoder_prediction int64 [T_dec, B])``.
    """
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    xe = np.asarray(encoder_inputs, dtype=dtype)[:, :, None]      # tf.expand_dims(., 2)  :283
    xd = np.asarray(decoder_inputs, dtype=dtype)[:, :, None]      # :285
    _, enc_state = dynamic_rnn(xe, p["enc_kernel"], p["enc_bias"])                      # :292-297
    dec_out, _ = dynamic_rnn(xd, p["dec_kernel"], p["dec_bias"], state=enc_state)       # :305-312
    logits = np.matmul(dec_out, p["proj_w"].reshape(-1, 1)) + p["proj_b"].reshape(1)   # :315-318
    prediction = np.argmax(logits, axis=2).astype(np.int64)                             # :321 (always 0)
    return logits[:, :, 0], prediction


def hard_bits(logits, block_length):
    """NOT in the reference (whose argmax over one logit is constantly 0): the sign decision a
    network trained towards +-1 targets (src/rnn_viterbi.py:324-326) would take, first
    ``block_length`` decoder steps, as [B, block_length] uint8.  Kept beside the literal output so
    the HIP kernel's optional bit/counter outputs have a checker."""
    return (np.asarray(logits)[:block_length].T > 0).astype(np.uint8)


def error_counters(bits, info_bits01):
    """{bit_errors, frame_errors, total_bits, total_frames} of hard decisions against 0/1 info bits."""
    diff = np.asarray(bits, np.uint8) != np.asarray(info_bits01, np.uint8)
    return np.array([int(diff.sum()), int(diff.any(axis=1).sum()), diff.size, diff.shape[0]], np.int64)
