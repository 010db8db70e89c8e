"""Host-side mirror of the reference's ``src/fnn_viterbi_seq_predict.py``: the
sliding-window ("sequential prediction") decoder.  One long (133,171)-coded
stream is received; a window of ``window_length`` coded symbols slides over it
two symbols at a time and a deep tanh network (200 -> 500 x4 -> 2) predicts the
information bit at the start of the window.

The forward pass runs on the MI355X through the layer-at-a-time HIP path
(``decoder.WideMLP``, C ABI ``nnv_wide_*``): layer 1 gathers window i straight
from the stream, so the ``[instances, window]`` matrix the reference fills row
by row in Python (src/fnn_viterbi_seq_predict.py:507-519,592-605) is never
built.  Dataset generation is numpy with the reference's RNG consumption
(golden vectors in tests/golden).

Reference behaviour kept on purpose: the noisy stream is
written back into the **int32** array the encoder produced, so every received
soft value is truncated toward zero before the network sees it
(src/fnn_viterbi_seq_predict.py:243-250,277-284).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import fnn_viterbi as _fv
from .fnn_viterbi import encode_133171, modulate_awgn  # noqa: F401

_W_KEYS = ("h1", "h2", "h3", "h4", "out")
_B_KEYS = ("b1", "b2", "b3", "b4", "out")
_VAR_W = {"h1": "weights_h1", "h2": "weights_h2", "h3": "weights_h3", "h4": "weights_h4",
          "out": "weights_out"}
_VAR_B = {"b1": "bias_h1", "b2": "bias_h2", "b3": "bias_h3", "b4": "bias_h4", "out": "bias_out"}


class SequentialDecoderConfig:
    """Same keyword arguments and derived attributes as the reference's
    ``SequentialDecoderConfig`` (src/fnn_viterbi_seq_predict.py:34-126)."""

    def __init__(self, total_train_length_info: int = 1000100, total_test_length_info: int = 1000100,
                 window_length: int = 200, n_hidden_1: int = 500, n_hidden_2: int = 500,
                 n_hidden_3: int = 500, n_hidden_4: int = 500, learning_rate: float = 0.0004,
                 training_epochs: int = 50, batch_size: int = 100, beta_regularization: float = 0.00000,
                 snr_db_range: Tuple[int, int] = (-2, 9), test_snr_index: int = 1,
                 model_save_path: str = "./models/seq_predict_model",
                 model_load_path: Optional[str] = None, test_old_model: bool = False):
        self.total_train_length_info = total_train_length_info
        self.total_train_length_coded = 2 * total_train_length_info
        self.total_test_length_info = total_test_length_info
        self.total_test_length_coded = 2 * total_test_length_info
        self.window_length = window_length
        self.training_size = 1
        self.test_size = 1
        self.total_training_instances = int((self.total_train_length_coded - window_length) * 0.5)
        self.total_test_instances = int((self.total_test_length_coded - window_length) * 0.5)
        self.n_input = window_length
        self.n_hidden_1, self.n_hidden_2 = n_hidden_1, n_hidden_2
        self.n_hidden_3, self.n_hidden_4 = n_hidden_3, n_hidden_4
        self.n_output = 2
        self.learning_rate = learning_rate
        self.training_epochs = training_epochs
        self.batch_size = batch_size
        self.beta_regularization = beta_regularization
        self.display_step = 1
        snr_db_vec = np.arange(snr_db_range[0], snr_db_range[1])
        self.snr_db_vec = snr_db_vec
        self.snr_vec = 10 ** (snr_db_vec / 10.0)
        self.n0_vec = 1.0 / self.snr_vec
        self.sigma_vec = np.sqrt(self.n0_vec * 0.5)
        self.test_snr_index = test_snr_index
        self.model_save_path = model_save_path
        self.model_load_path = model_load_path
        self.test_old_model = test_old_model


# ==============================================================================
# streams  (reference: src/fnn_viterbi_seq_predict.py:133-286)
# ==============================================================================


def generate_info_sequence(block_length: int) -> np.ndarray:
    """``np.random.randint(2, size=block_length)`` (src/fnn_viterbi_seq_predict.py:133-143)."""
    return np.random.randint(2, size=block_length)


def _noisy_streams_truncated(info_seq: np.ndarray, sigma_for_row) -> np.ndarray:
    """Encode every row, then overwrite each int32 row with BPSK + AWGN: the float64
    values are truncated toward zero by the int32 destination, exactly as in the
    reference."""
    encoded = _fv.encode_133171_batch(info_seq)                       # int32 [rows, 2L]
    for i in range(encoded.shape[0]):
        encoded[i, :] = modulate_awgn(encoded[i, :], sigma_for_row(i))
    return encoded


def generate_training_data(total_length_info: int, training_size: int,
                           sigma_vec: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(info float32 [size, L], coded int32 [size, 2L]); each row gets ONE random SNR out of
    ``sigma_vec``, drawn after all info bits (src/fnn_viterbi_seq_predict.py:216-252)."""
    info_seq = np.random.randint(2, size=(training_size, total_length_info)).astype(np.float32)
    return info_seq, _noisy_streams_truncated(
        info_seq, lambda i: sigma_vec[np.random.randint(len(sigma_vec))])


def generate_test_data(total_length_info: int, test_size: int, sigma_vec: np.ndarray,
                       noise_index: int) -> Tuple[np.ndarray, np.ndarray]:
    """(info int64 [size, L], coded int32 [size, 2L]) at the fixed SNR ``sigma_vec[noise_index]``
    (src/fnn_viterbi_seq_predict.py:255-286)."""
    info_seq = np.random.randint(2, size=(test_size, total_length_info))
    test_sigma = sigma_vec[noise_index]
    return info_seq, _noisy_streams_truncated(info_seq, lambda i: test_sigma)


def get_single_instance(window_length: int, coded_seq: np.ndarray, info_seq: np.ndarray,
                        offset: int) -> Tuple[np.ndarray, np.ndarray]:
    """Window ``coded_seq[0, offset : offset + window_length]`` and the one-hot label of the
    information bit ``info_seq[0, offset // 2]`` (src/fnn_viterbi_seq_predict.py:293-336)."""
    batch_x = np.asarray(coded_seq)[0, offset:offset + window_length]
    batch_y = np.zeros(2, dtype=np.float32)
    batch_y[int(np.asarray(info_seq)[0, int(offset * 0.5)])] = 1.0
    return batch_x, batch_y


def extract_windows(window_length: int, coded_seq: np.ndarray, first_offset: int,
                    count: int) -> np.ndarray:
    """``count`` consecutive instances starting at ``first_offset`` (step 2) as one float32
    [count, window] array: the vectorised form of the reference's per-row loops."""
    row = np.asarray(coded_seq)[0]
    view = np.lib.stride_tricks.sliding_window_view(row, window_length)[first_offset::2][:count]
    return np.ascontiguousarray(view, dtype=np.float32)


# ==============================================================================
# model  (reference: src/fnn_viterbi_seq_predict.py:343-414)
# ==============================================================================


def layer_dims(config: SequentialDecoderConfig) -> list:
    return [config.n_input, config.n_hidden_1, config.n_hidden_2, config.n_hidden_3,
            config.n_hidden_4, config.n_output]


def initialize_weights(config: SequentialDecoderConfig, rng=None) -> Tuple[Dict, Dict]:
    """Truncated normal, stddev sqrt(2/fan_in), zero biases; keys h1..h4,out / b1..b4,out
    (src/fnn_viterbi_seq_predict.py:360-404)."""
    dims = layer_dims(config)
    weights = {k: _fv.truncated_normal((dims[i], dims[i + 1]), math.sqrt(2.0 / dims[i]), rng)
               for i, k in enumerate(_W_KEYS)}
    biases = {k: np.zeros(dims[i + 1], dtype=np.float32) for i, k in enumerate(_B_KEYS)}
    return weights, biases


def weights_as_lists(weights: Dict, biases: Dict):
    return [weights[k] for k in _W_KEYS], [biases[k] for k in _B_KEYS]


def build_network(x, config: SequentialDecoderConfig, weights: Optional[Dict] = None,
                  biases: Optional[Dict] = None):
    """(output, weights, biases) like the reference (src/fnn_viterbi_seq_predict.py:343-414).
    Fresh variables are initialised unless given; ``output`` is the tanh'd class layer
    [N, 2] for the rows of ``x`` computed on the GPU (None when ``x`` is None)."""
    if weights is None or biases is None:
        weights, biases = initialize_weights(config)
    output = None
    if x is not None:
        import torch
        xd = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).cuda()
        output = _decoder_for(weights, biases).decode(xd, want_act=True)["act"].cpu().numpy()
    return output, weights, biases


def _decoder_for(weights: Dict, biases: Dict, device=None):
    """The packed sliding-window network, kept by ``decoder.cached_network`` while calls keep coming with the same
    weights (3.4 MB packed twice -- layered + K6 -- and uploaded ONCE, not around every call).  Never closed here."""
    import torch
    from .decoder import WideMLP, _require_cuda, cached_network
    _require_cuda()
    Ws, bs = weights_as_lists(weights, biases)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        return cached_network("tanh", list(Ws) + list(bs), lambda: WideMLP(Ws, bs, device=dev))


def predict_stream(coded_seq: np.ndarray, n_instances: int, weights: Dict, biases: Dict) -> np.ndarray:
    """``predictions_tensor.eval({x: test_data})`` for test_data = the first ``n_instances``
    windows of ``coded_seq`` (src/fnn_viterbi_seq_predict.py:507-524,592-611), int64."""
    import torch
    stream = torch.from_numpy(np.ascontiguousarray(np.asarray(coded_seq)[0], dtype=np.float32)).cuda()
    return _decoder_for(weights, biases).decode_stream(stream, n_instances, want_cls64=True)["cls64"].cpu().numpy()


def generate_test_stream_on_device(n_info_bits: int, sigma: float, seed: int, device=None):
    """The test stream of ``generate_test_data`` (:255-286) for LONG streams, made on the GPU: ``n_info_bits`` random
    bits (rounded up to a multiple of 8) -> (133,171) from zero state as ONE stream -> BPSK + AWGN -> truncated toward
    zero like the reference's int32 destination.  Returns ``(info uint8 [n] , coded float32 [2n])`` CUDA
    tensors.  Philox-keyed by (seed, position): not the reference's numpy draws (a device generator cannot be), the
    same contract as ``decoder.generate_awgn_blocks(..., continuous=True)``, whose rows ARE this stream."""
    import torch
    from .decoder import generate_awgn_blocks
    rows = (int(n_info_bits) + 7) // 8
    x, labels = generate_awgn_blocks("133171", 8, rows, float(sigma), int(seed), rows_per_block=rows,
                                     device=device, continuous=True)
    shifts = torch.arange(7, -1, -1, dtype=torch.int32, device=x.device)
    info = ((labels[:, None] >> shifts[None, :]) & 1).to(torch.uint8).reshape(-1)
    return info, torch.trunc(x.reshape(-1))


def evaluate_stream_on_device(weights: Dict, biases: Dict, n_info_bits: int, sigma: float, seed: int,
                              window_length: Optional[int] = None, device=None) -> Dict:
    """Generate a test stream on the GPU and decode every full window of it (window i = coded[2i : 2i + window] decides
    info bit i, :293-336) in one persistent launch; returns the error count and rate over the decodable bits."""
    import torch
    Ws, bs = weights_as_lists(weights, biases)
    window = int(np.asarray(Ws[0]).shape[0]) if window_length is None else int(window_length)
    info, coded = generate_test_stream_on_device(n_info_bits, sigma, seed, device=device)
    n_instances = (coded.numel() - window) // 2 + 1
    if n_instances < 1:
        raise ValueError("the stream is shorter than one window")
    pred = _decoder_for(weights, biases, device=coded.device).decode_stream(coded, n_instances)["cls32"]
    errors = int((pred != info[:n_instances].to(torch.int32)).sum().item())
    return {"decoded_bits": n_instances, "bit_errors": errors, "ber": errors / n_instances}


def _ckpt(path: str) -> str:
    return path if path.endswith(".npz") else path + ".npz"


def save_model(path: str, weights: Dict, biases: Dict) -> str:
    path = _ckpt(path)
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    arrays = {_VAR_W[k]: np.asarray(v, np.float32) for k, v in weights.items()}
    arrays.update({_VAR_B[k]: np.asarray(v, np.float32) for k, v in biases.items()})
    np.savez(path, **arrays)
    return path


def load_model(path: str) -> Tuple[Dict, Dict]:
    data = np.load(_ckpt(path), allow_pickle=False)
    return ({k: data[v] for k, v in _VAR_W.items()}, {k: data[v] for k, v in _VAR_B.items()})


# ==============================================================================
# training + evaluation  (reference: src/fnn_viterbi_seq_predict.py:421-621)
# ==============================================================================


def _train(config: SequentialDecoderConfig, train_info, train_coded, weights, biases):
    """Adam on mean sigmoid cross-entropy taken on the tanh outputs + beta*(l2(h1)+l2(h2)),
    batches of consecutive windows walking the stream, offset reset every epoch
    (src/fnn_viterbi_seq_predict.py:470-487,541-577).  Plain PyTorch autograd."""
    import torch
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    from .training import NativeDenseTrainer
    bs, win = int(config.batch_size), int(config.window_length)
    total_batch = int(config.total_training_instances / bs)
    if total_batch > 0 and NativeDenseTrainer.eligible([weights[k] for k in _W_KEYS], config.n_output):
        # K8 (csrc/nnv_dense_train.hip): the step as 11 hand-written launches; window i = stream[2 i : 2 i + win] is
        # addressed inside the staging kernel (row stride 2), so no [instances, win] matrix exists
        nt = NativeDenseTrainer([weights[k] for k in _W_KEYS], [biases[k] for k in _B_KEYS], "tanh", config.learning_rate,
                                config.beta_regularization, bs, reg_layers=(0, 1))
        stream = torch.as_tensor(np.asarray(train_coded)[0].astype(np.float32), device=nt.device).contiguous()
        lab = torch.as_tensor(np.asarray(train_info)[0, :total_batch * bs].astype(np.int32), device=nt.device).contiguous()
        for epoch in range(int(config.training_epochs)):
            nt.steps(stream, lab, total_batch, row_stride=2, step_stride=2 * bs)
            if epoch % config.display_step == 0:
                print(f"Epoch: {epoch + 1:04d} | cost = {nt.export(want_cost_sum=True)[3] / total_batch:.9f}")
        Ws, Bs, _ = nt.export()
        nt.close()
        return dict(zip(_W_KEYS, Ws)), dict(zip(_B_KEYS, Bs))
    W = [torch.tensor(weights[k], dtype=torch.float32, device=dev, requires_grad=True) for k in _W_KEYS]
    B = [torch.tensor(biases[k], dtype=torch.float32, device=dev, requires_grad=True) for k in _B_KEYS]
    stream = torch.as_tensor(np.asarray(train_coded)[0].astype(np.float32), device=dev)
    bits = torch.as_tensor(np.asarray(train_info)[0].astype(np.int64), device=dev)
    from .training import GraphedStep, TFAdam
    opt = TFAdam(W + B, lr=config.learning_rate)      # tf.train.AdamOptimizer (src/fnn_viterbi_seq_predict.py:488), TF-1's update rule
    windows = stream.unfold(0, win, 2)                      # window i = stream[2i : 2i + win]

    def step(xb, bb):
        yb = torch.nn.functional.one_hot(bb, 2).to(torch.float32)
        h = xb
        for l in range(4):
            h = torch.tanh(h @ W[l] + B[l])
        pred = torch.tanh(h @ W[4] + B[4])
        ce = torch.nn.functional.binary_cross_entropy_with_logits(pred, yb)
        cost = ce + config.beta_regularization * 0.5 * ((W[0] ** 2).sum() + (W[1] ** 2).sum())
        opt.zero_grad(set_to_none=False)
        cost.backward()
        opt.step()
        return cost.detach()

    run = GraphedStep(step, W + B, opt, [windows[:bs], bits[:bs]]) if total_batch > 0 else step
    for epoch in range(int(config.training_epochs)):
        avg_cost = torch.zeros((), device=dev)
        for i in range(total_batch):
            avg_cost += run(windows[i * bs:(i + 1) * bs], bits[i * bs:(i + 1) * bs]) / max(total_batch, 1)
        if epoch % config.display_step == 0:
            print(f"Epoch: {epoch + 1:04d} | cost = {float(avg_cost):.9f}")
    return ({k: w.detach().cpu().numpy().copy() for k, w in zip(_W_KEYS, W)},
            {k: b.detach().cpu().numpy().copy() for k, b in zip(_B_KEYS, B)})


def train_and_evaluate(config: SequentialDecoderConfig) -> Dict:
    """The reference's pipeline with its printed lines (src/fnn_viterbi_seq_predict.py:421-621).
    Returns a dict with the predictions, error rate and the model (the reference returns None)."""
    print("=" * 70)
    print("Sequential Prediction Neural Decoder")
    print("=" * 70)
    print("\nConfiguration:")
    print(f"  Training stream length: {config.total_train_length_info} bits")
    print(f"  Test stream length: {config.total_test_length_info} bits")
    print(f"  Window length: {config.window_length} encoded bits")
    print(f"  Training instances: {config.total_training_instances}")
    print(f"  Test instances: {config.total_test_instances}")
    print(f"  Test SNR: {config.snr_db_vec[config.test_snr_index]} dB")
    print("\nGenerating data streams...")
    train_info, train_coded = generate_training_data(config.total_train_length_info,
                                                     config.training_size, config.sigma_vec)
    test_info, test_coded = generate_test_data(config.total_test_length_info, config.test_size,
                                               config.sigma_vec, config.test_snr_index)
    print(f"Training stream shape: {train_coded.shape}")
    print(f"Test stream shape: {test_coded.shape}")
    if config.test_old_model:
        print("\nLoading existing model...")
        if not config.model_load_path:
            raise RuntimeError("test_old_model=True needs model_load_path: no trained variables")
        weights, biases = load_model(config.model_load_path)
        print(f"Model loaded from {config.model_load_path}")
        print("\nPreparing test data...")
        print("Evaluating...")
    else:
        weights, biases = initialize_weights(config)
        print("\nStarting training...")
        weights, biases = _train(config, train_info, train_coded, weights, biases)
        print("Optimization Finished!")
        save_model(config.model_save_path, weights, biases)
        print(f"Model saved to {config.model_save_path}")
        print("\nPreparing test data...")
        print("Evaluating on test set...")
    predictions = predict_stream(test_coded, config.total_test_instances, weights, biases)
    labels = np.asarray(test_info)[0, :config.total_test_instances].astype(np.int64)
    errors = int(np.sum(predictions != labels))
    error_rate = errors / len(predictions)
    print(f"\nTest Error Rate: {error_rate:.6f}")
    print(f"Test Accuracy: {1 - error_rate:.6f}")
    if not config.test_old_model:
        print(f"SNR: {config.snr_db_vec[config.test_snr_index]} dB")
    print("\n" + "=" * 70)
    print("Decoding complete!")
    print("=" * 70)
    return {"predictions": predictions, "error_rate": error_rate, "weights": weights, "biases": biases,
            "test_info": test_info, "test_coded": test_coded}


def main() -> None:
    train_and_evaluate(SequentialDecoderConfig())


if __name__ == "__main__":
    main()
