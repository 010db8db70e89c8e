"""Host-side mirror of the reference's ``src/rnn_viterbi.py``: the recurrent
(LSTM sequence-to-sequence) decoder.  SURVEY.md section 8f row 4 / BASELINE configs[4].

The forward pass (LSTM encoder over the received symbols -> LSTM decoder started
from the encoder's final state -> one linear logit per decoder step) runs on the
MI355X through ``decoder.Seq2SeqLSTM`` (C ABI ``nnv_seq2seq_*``); dataset
generation is numpy with the reference's RNG consumption; training is plain
PyTorch autograd (not a hot path).

The reference script cannot be imported or run: ``Optional`` is never
imported (:35), 3-D arrays are fed into 2-D placeholders (:361-392 vs :263-279).
What is mirrored here is the graph and the data flow AS WRITTEN, with the
behaviour that follows from it kept on purpose:
  * the network is an LSTM (BASELINE.json says GRU; the code says ``LSTMCell`` :289,302);
  * ``vocab_size = 1`` (:71): ``decoder_prediction = tf.argmax(logits, 2)`` (:321) is
    identically 0, so the BER the script prints is the fraction of 1 bits in the test
    information (:479-496).  ``train_seq2seq`` reproduces that number; the sign
    decision ``logit > 0`` the MSE-towards-+-1 training (:324-326) implies is offered
    beside it as ``run_seq2seq(...)["hard_bits"]`` and is labelled as an extension;
  * noisy codewords are written back into the int32 array the encoder produced, so the
    soft values are truncated toward zero (:201-208,216-222);
  * the decoder is teacher-forced with the TRUE information bits even at test time
    (``prepare_batch`` builds decoder_inputs from ``info_batch`` :378-386).
The feeds are passed time-major ``[T, B]``, the declared placeholder shape; the extra
trailing axis ``prepare_batch`` adds (the reference's shape bug) is accepted and dropped.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import fnn_viterbi as _fv
from .fnn_viterbi import encode_133171, modulate_awgn  # noqa: F401  (same text as src/rnn_viterbi.py:99-166)

_PARAM_KEYS = ("enc_kernel", "enc_bias", "dec_kernel", "dec_bias", "proj_w", "proj_b")


class RNNDecoderConfig:
    """Same keyword arguments and derived attributes as the reference's
    ``RNNDecoderConfig`` (src/rnn_viterbi.py:28-92)."""

    def __init__(self, block_length: int = 100, encoder_hidden_units: int = 20,
                 decoder_hidden_units: Optional[int] = None, learning_rate: float = 0.01,
                 training_epochs: int = 6, batch_size: int = 100, beta_regularization: float = 0.001,
                 training_size: int = 100000, test_size: int = 1000,
                 snr_db_range: Tuple[int, int] = (0, 9), test_snr_index: int = 2):
        self.block_length = block_length
        self.encoded_length = 2 * block_length
        self.encoder_hidden_units = encoder_hidden_units
        self.decoder_hidden_units = decoder_hidden_units if decoder_hidden_units else encoder_hidden_units
        self.vocab_size = 1
        self.learning_rate = learning_rate
        self.training_epochs = training_epochs
        self.batch_size = batch_size
        self.beta_regularization = beta_regularization
        self.display_step = 1
        self.training_size = training_size
        self.test_size = test_size
        snr_db_vec = np.arange(snr_db_range[0], snr_db_range[1])
        self.snr_db_vec = snr_db_vec
        self.snr_vec = 10 ** (snr_db_vec / 10.0)
        self.n0_vec = 1.0 / self.snr_vec
        self.sigma_vec = np.sqrt(self.n0_vec * 0.5)
        self.test_snr_index = test_snr_index
        self.PAD = 0
        self.EOS = 1


# ==============================================================================
# data generation  (reference: src/rnn_viterbi.py:173-229)
# ==============================================================================


def generate_input_vectors(length: int, number: int) -> np.ndarray:
    """``np.random.randint(2, size=(number, length))`` (src/rnn_viterbi.py:173-184)."""
    return np.random.randint(2, size=(number, length))


def _noisy_truncated(info: np.ndarray, sigma: float) -> np.ndarray:
    """encode every row, BPSK + AWGN row by row, stored into the encoder's int32 array
    (truncation toward zero).  One [rows, 2L] normal draw = the reference's per-row draws
    (numpy's legacy generator yields the same stream)."""
    coded = _fv.encode_133171_batch(info)
    sym = coded.astype(np.float32)
    sym[sym == 0] = -1
    noisy = sym + np.random.normal(0, sigma, sym.shape)
    out = np.empty_like(coded)
    out[...] = noisy          # float64 -> int32: toward zero, as ``train_encoded[i, :] = ...`` does
    return out


def _pm1(info: np.ndarray) -> np.ndarray:
    a = info.astype(np.float32)
    a[a == 0] = -1
    return a


def generate_datasets(config: RNNDecoderConfig):
    """-> (train_info f32 +-1 [N, L], train_encoded int32 [N, 2L], test_info, test_encoded),
    RNG order as in the reference: training info, training noise, test info, test noise; both
    sets at ``sigma_vec[test_snr_index]`` (src/rnn_viterbi.py:187-229)."""
    sigma = config.sigma_vec[config.test_snr_index]
    train_info = generate_input_vectors(config.block_length, config.training_size)
    train_encoded = _noisy_truncated(train_info, sigma)
    test_info = generate_input_vectors(config.block_length, config.test_size)
    test_encoded = _noisy_truncated(test_info, sigma)
    return _pm1(train_info), train_encoded, _pm1(test_info), test_encoded


# ==============================================================================
# model  (reference: src/rnn_viterbi.py:235-340)
# ==============================================================================


def initialize_parameters(config: RNNDecoderConfig, rng=None) -> Dict[str, np.ndarray]:
    """The variables ``build_seq2seq_model`` creates, with TF-1.x's defaults for them:
    ``LSTMCell`` kernel [1 + H, 4H] Glorot-uniform (get_variable's default initializer), bias
    zeros; ``layers.linear`` weights [H, 1] Xavier-uniform, bias zeros."""
    if config.decoder_hidden_units != config.encoder_hidden_units:
        raise ValueError("the decoder starts from the encoder's state (src/rnn_viterbi.py:308): "
                         "decoder_hidden_units must equal encoder_hidden_units")
    rng = np.random if rng is None else rng
    H = int(config.encoder_hidden_units)

    def glorot(shape):
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    return {
        "enc_kernel": glorot((1 + H, 4 * H)), "enc_bias": np.zeros(4 * H, np.float32),
        "dec_kernel": glorot((1 + H, 4 * H)), "dec_bias": np.zeros(4 * H, np.float32),
        "proj_w": glorot((H, config.vocab_size)), "proj_b": np.zeros(config.vocab_size, np.float32),
    }


def build_seq2seq_model(config: RNNDecoderConfig, params: Optional[Dict[str, np.ndarray]] = None,
                        rng=None) -> Dict:
    """Counterpart of the reference's graph builder.  Returns a model dict holding the config
    and the parameters (the reference returns placeholder/tensor handles); ``run_seq2seq`` and
    ``train_seq2seq`` act on it."""
    p = initialize_parameters(config, rng) if params is None else {k: np.asarray(params[k], np.float32)
                                                                   for k in _PARAM_KEYS}
    return {"config": config, "params": p, "forget_bias": 1.0}


def prepare_batch(info_batch: np.ndarray, encoded_batch: np.ndarray, config: RNNDecoderConfig) -> Dict:
    """The three feeds, exactly as the reference builds them (src/rnn_viterbi.py:343-392),
    trailing singleton axis included: encoder_inputs [2L, B, 1] (transposed codewords),
    decoder_targets = [info | EOS | zeros], decoder_inputs = [EOS | info | zeros]."""
    batch_size = len(info_batch)
    encoder_inputs = np.expand_dims(np.transpose(encoded_batch), axis=2)
    decoder_targets = np.concatenate([
        info_batch,
        np.tile(np.concatenate([[config.EOS], np.zeros(config.block_length - 1)]), (batch_size, 1)),
    ], axis=1)
    decoder_targets = np.expand_dims(np.transpose(decoder_targets), axis=2)
    decoder_inputs = np.concatenate([np.tile([config.EOS], (batch_size, 1)), info_batch], axis=1)
    decoder_inputs = np.concatenate([decoder_inputs, np.zeros((batch_size, config.block_length - 1))], axis=1)
    decoder_inputs = np.expand_dims(np.transpose(decoder_inputs), axis=2)
    return {"encoder_inputs": encoder_inputs, "decoder_inputs": decoder_inputs,
            "decoder_targets": decoder_targets}


def _time_major(feed: np.ndarray) -> np.ndarray:
    a = np.asarray(feed)
    if a.ndim == 3 and a.shape[2] == 1:      # prepare_batch's extra axis
        a = a[:, :, 0]
    if a.ndim != 2:
        raise ValueError("feeds are time-major [T, B] (src/rnn_viterbi.py:263-279)")
    return np.ascontiguousarray(a, dtype=np.float32)


def run_seq2seq(model: Dict, feeds: Dict, n_bits: Optional[int] = None, info_bits01=None,
                device=None, precision: Optional[str] = None) -> Dict:
    """``sess.run(model['decoder_prediction'], feed_dict)`` (src/rnn_viterbi.py:471-474) on the
    MI355X.  Returns ``decoder_logits`` float32 [T_dec, B], ``decoder_prediction`` int64 [T_dec, B]
    (argmax over the size-1 vocabulary axis: zeros), and, as an extension, ``hard_bits`` uint8
    [B, n_bits] (logit > 0) with error ``counters`` against ``info_bits01`` [B, n_bits] if given."""
    import torch
    from .decoder import Seq2SeqLSTM, _require_cuda, cached_network
    _require_cuda()                      # no CPU path: fails loudly (NNViterbiError) without an MI355X
    cfg = model["config"]
    enc = _time_major(feeds["encoder_inputs"])
    dec = _time_major(feeds["decoder_inputs"])
    n_bits = cfg.block_length if n_bits is None else int(n_bits)
    keys = ("enc_kernel", "enc_bias", "dec_kernel", "dec_bias", "proj_w", "proj_b")
    fb = float(model.get("forget_bias", 1.0))
    dev0 = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev0):     # kept while calls keep coming with the same parameters: one prepack, never closed here
        net = cached_network("seq2seq", [model["params"][k] for k in keys],
                             lambda: Seq2SeqLSTM(model["params"], forget_bias=fb, device=dev0, precision=precision),
                             extra=(fb, precision))     # "f16x3": hi/lo f16 operands, ~2x, same 5e-5 tolerance
    dev = net.device
    info = None
    if info_bits01 is not None:
        info = torch.from_numpy(np.ascontiguousarray(np.asarray(info_bits01, np.uint8).T)).to(dev)
    r = net.forward(torch.from_numpy(enc).to(dev), torch.from_numpy(dec).to(dev),
                    info_bits=info, n_bits=n_bits, want_bits=True)
    torch.cuda.synchronize(dev)
    logits = r["logits"].cpu().numpy()
    out = {
        "decoder_logits": logits,
        "decoder_prediction": np.zeros(logits.shape, np.int64),   # argmax over one logit (:321)
        "hard_bits": np.ascontiguousarray(r["bits"].cpu().numpy().T),
    }
    if r["counters"] is not None:
        out["counters"] = r["counters"].cpu().numpy()
    return out


# ==============================================================================
# training and evaluation  (reference: src/rnn_viterbi.py:399-505)
# ==============================================================================


def _torch_forward(P, enc, dec, forget_bias):
    """The same graph in torch ops, for autograd (time-major [T, B] inputs)."""
    import torch
    B = enc.shape[1]
    H = P["enc_bias"].shape[0] // 4
    c = enc.new_zeros((B, H))
    m = enc.new_zeros((B, H))

    def cell(x_t, c, m, kernel, bias):
        z = torch.cat([x_t[:, None], m], dim=1) @ kernel + bias
        i, j, f, o = z[:, 0:H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:4 * H]
        c = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
        return c, torch.sigmoid(o) * torch.tanh(c)

    for t in range(enc.shape[0]):
        c, m = cell(enc[t], c, m, P["enc_kernel"], P["enc_bias"])
    outs = []
    for t in range(dec.shape[0]):
        c, m = cell(dec[t], c, m, P["dec_kernel"], P["dec_bias"])
        outs.append(m)
    return torch.stack(outs) @ P["proj_w"].reshape(H, 1) + P["proj_b"]        # [T, B, 1]


def train_seq2seq(config: RNNDecoderConfig, train_info: np.ndarray, train_encoded: np.ndarray,
                  test_info: np.ndarray, test_encoded: np.ndarray, model: Optional[Dict] = None,
                  verbose: bool = True) -> Tuple[np.ndarray, float]:
    """Train (Adam on the mean squared difference between decoder_targets and the logits,
    contiguous batches cycling over the training set, src/rnn_viterbi.py:409-441) and evaluate
    on the test set through the HIP forward.  Returns ``(predictions [test_size, L], ber)`` with
    the reference's literal semantics: predictions are the argmax over one logit (zeros), ber is
    their disagreement with the information bits.  The trained parameters are left in
    ``model["params"]`` when a model dict is passed in."""
    import torch
    model = build_seq2seq_model(config) if model is None else model
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    P = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True)
         for k, v in model["params"].items()}
    from .training import GraphedStep, TFAdam
    opt = TFAdam(list(P.values()), lr=config.learning_rate)   # tf.train.AdamOptimizer (src/rnn_viterbi.py:316), TF-1's update rule
    batches_per_epoch = int(config.training_size / config.batch_size)
    max_batches = batches_per_epoch * config.training_epochs
    loss_history = []
    forget_bias = float(model.get("forget_bias", 1.0))

    from .training import lstm_native_eligible, lstm_seq2seq_logits_native
    # K9 (csrc/nnv_lstm_train.hip): each recurrence one launch forward and one backward instead of ~40 kernels per cell
    forward = lstm_seq2seq_logits_native if (dev.type == "cuda" and lstm_native_eligible(P)) else _torch_forward

    def step(enc, dec, tgt):
        loss = ((tgt - forward(P, enc, dec, forget_bias)) ** 2).mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss.detach()

    def feeds_of(lo):
        f = prepare_batch(train_info[lo:lo + config.batch_size], train_encoded[lo:lo + config.batch_size], config)
        return (torch.as_tensor(_time_major(f["encoder_inputs"]), device=dev),
                torch.as_tensor(_time_major(f["decoder_inputs"]), device=dev),
                torch.as_tensor(np.asarray(f["decoder_targets"], np.float32), device=dev))

    # one batch step = 4L cell evaluations = thousands of tiny kernels: captured once, replayed per batch
    run = GraphedStep(step, list(P.values()), opt, feeds_of(0), verbose) if max_batches > 0 else step
    if verbose:
        print(f"Training for {config.training_epochs} epochs ({max_batches} batches)...")
    for batch_idx in range(max_batches):
        start = (batch_idx % batches_per_epoch) * config.batch_size
        info_batch = train_info[start:start + config.batch_size]
        encoded_batch = train_encoded[start:start + config.batch_size]
        loss_history.append(float(run(*feeds_of(start))))
        if verbose and batch_idx % batches_per_epoch == 0:
            epoch = batch_idx // batches_per_epoch
            print(f"Epoch {epoch + 1}/{config.training_epochs} | Loss: {loss_history[-1]:.6f}")
            if epoch == 0 or (epoch + 1) % 2 == 0:
                print(f"  Sample input:  {encoded_batch[0, :10]}")
                print(f"  Sample target: {info_batch[0, :10]}")
                print(f"  Sample pred:   {np.zeros(10, np.int64)}")
    model["params"] = {k: v.detach().cpu().numpy().copy() for k, v in P.items()}
    model["loss_history"] = loss_history
    if verbose:
        print("Training finished!")
        print("\nEvaluating on test set...")
    out = run_seq2seq(model, prepare_batch(test_info, test_encoded, config))
    predictions = np.transpose(out["decoder_prediction"])[:, :config.block_length]
    test_info_binary = test_info.copy()
    test_info_binary[test_info_binary == -1] = 0
    bit_errors = int(np.logical_xor(predictions, test_info_binary).sum())
    total_bits = config.test_size * config.block_length
    ber = bit_errors / total_bits
    model["hard_bits"] = out["hard_bits"]
    if verbose:
        print(f"Bit Error Rate: {ber:.6f}")
        print(f"Bit Errors: {bit_errors}/{total_bits}")
    try:                                     # the reference saves a loss plot (:498-505)
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.figure(figsize=(10, 6))
        plt.plot(loss_history)
        plt.xlabel("Batch")
        plt.ylabel("Loss")
        plt.title("Training Loss History")
        plt.grid(True)
        plt.savefig("rnn_training_loss.png")
        if verbose:
            print("Loss plot saved to rnn_training_loss.png")
    except ImportError:
        if verbose:
            print("matplotlib is not installed: loss plot skipped")
    return predictions, ber


def save_model(path: str, model: Dict) -> str:
    path = path if path.endswith(".npz") else path + ".npz"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, **{k: np.asarray(model["params"][k], np.float32) for k in _PARAM_KEYS})
    return path


def load_model(path: str, config: RNNDecoderConfig) -> Dict:
    path = path if path.endswith(".npz") else path + ".npz"
    with np.load(path, allow_pickle=False) as f:
        return build_seq2seq_model(config, params={k: f[k] for k in _PARAM_KEYS})


def main() -> None:
    """src/rnn_viterbi.py:512-557."""
    print("=" * 70)
    print("RNN-Based Viterbi Decoder (Sequence-to-Sequence)")
    print("=" * 70)
    config = RNNDecoderConfig(block_length=100, encoder_hidden_units=20, learning_rate=0.01,
                              training_epochs=6, batch_size=100, training_size=100000,
                              test_size=1000, test_snr_index=2)
    print("\nConfiguration:")
    print(f"  Block Length: {config.block_length}")
    print(f"  Encoder Hidden Units: {config.encoder_hidden_units}")
    print(f"  Decoder Hidden Units: {config.decoder_hidden_units}")
    print(f"  Test SNR: {config.snr_db_vec[config.test_snr_index]} dB")
    print(f"  Training Samples: {config.training_size}")
    print(f"  Test Samples: {config.test_size}")
    print("\nGenerating datasets...")
    train_info, train_encoded, test_info, test_encoded = generate_datasets(config)
    print("\nTraining RNN decoder...")
    train_seq2seq(config, train_info, train_encoded, test_info, test_encoded)
    print("\n" + "=" * 70)
    print("Decoding complete!")
    print("=" * 70)


if __name__ == "__main__":
    main()
