"""Host-side mirror of the reference's ``src/fnn_viterbi.py`` surface.

Same names, argument meaning, return types, printed lines and error behaviour
as the reference for the feed-forward decoder; the arithmetic of the forward
pass runs in the HIP kernel (``decoder.PackedMLP``), never on the CPU, and the
Python per-row loops of the reference are replaced by vectorised numpy that is
tested equal to them (tests/test_host_api.py, golden vectors generated from
the reference itself).

Reference quirks that are part of the API and therefore kept:
  * ``encode_133171`` raises IndexError for fewer than 6 bits (:201);
  * ``encode_57`` emits (5,7) order, output index 1 is u[0]+u[1] and index 0
    is u[0] without the mod-2; IndexError for fewer than 2 bits (:142-145);
  * ``train_test_dataset`` always encodes with ``encode_133171`` (:352);
  * sigma = sqrt(0.5 / 10^(snr_db/10)) - Es/N0 per coded symbol (:89-92);
  * ``modulate_awgn`` returns float64 (float32 symbols + float64 noise, :244-251).
"""
from __future__ import annotations

import math
from random import randint
from typing import Dict, Optional, Tuple

import numpy as np

# ==============================================================================
# configuration  (reference: src/fnn_viterbi.py:27-92)
# ==============================================================================


class DecoderConfig:
    """Same keyword arguments and derived attributes as the reference's
    ``DecoderConfig`` (src/fnn_viterbi.py:27-92)."""

    def __init__(self, block_length: int = 7, rate: int = 2, snr_db: float = 2.0,
                 n_hidden_1: int = 128, n_hidden_2: int = 64, n_hidden_3: int = 32,
                 learning_rate: float = 0.0001, training_epochs: int = 200,
                 batch_size: int = 256, beta_regularization: float = 0.001,
                 samples_per_class_train: int = 500,
                 samples_per_class_test: int = 10000):
        self.block_length = block_length
        self.rate = rate
        self.all_classes = 2 ** block_length
        self.num_classes = self.all_classes
        self.n_input = rate * block_length
        self.n_hidden_1, self.n_hidden_2, self.n_hidden_3 = n_hidden_1, n_hidden_2, n_hidden_3
        self.n_classes = self.all_classes
        self.learning_rate = learning_rate
        self.training_epochs = training_epochs
        self.batch_size = batch_size
        self.beta_regularization = beta_regularization
        self.display_step = 1
        self.samples_per_class_train = samples_per_class_train
        self.samples_per_class_test = samples_per_class_test
        self.num_training_samples = self.num_classes * samples_per_class_train
        self.num_test_samples = self.num_classes * samples_per_class_test
        self.snr_db = snr_db
        self.snr = 10 ** (snr_db / 10.0)
        self.n0 = 1.0 / self.snr
        self.sigma = np.sqrt(self.n0 * 0.5)


# ==============================================================================
# encoders and channel  (reference: src/fnn_viterbi.py:99-253)
# ==============================================================================


def generate_info_sequence(block_length: int) -> np.ndarray:
    """Random 0/1 int32 sequence drawn bit by bit from ``random.randint``
    (src/fnn_viterbi.py:116-119), so ``random.seed`` reproduces the reference."""
    return np.fromiter((randint(0, 1) for _ in range(block_length)),
                       dtype=np.int32, count=block_length)


def _conv_parity(u: np.ndarray, taps) -> np.ndarray:
    """Parity of u[i - d] summed over the taps d (zero register at the start), uint8.  The parity of a sum of
    integers is the XOR of their low bits, so the shifted copies are XORed in place as bytes: no int64
    temporaries (at the bit-wise decoder's 300,000 x 518 rows they were 1.2 GB each and the whole cost)."""
    low = (np.asarray(u) & 1).astype(np.uint8)
    acc = np.zeros(low.shape, dtype=np.uint8)
    n = low.shape[-1]
    for d in taps:
        if d == 0:
            acc ^= low
        elif d < n:
            acc[..., d:] ^= low[..., :n - d]
    return acc


_TAPS_133 = (0, 2, 3, 5, 6)
_TAPS_171 = (0, 1, 2, 3, 6)


def encode_133171_batch(bits: np.ndarray) -> np.ndarray:
    """Rows of ``bits`` [..., L] encoded independently from zero state with the
    (133,171) rate-1/2 K=7 code; output [..., 2L] int32, pair i = (g133, g171).
    Vectorised twin of :func:`encode_133171`."""
    u = np.asarray(bits)
    if not np.issubdtype(u.dtype, np.integer):
        u = u.astype(np.int64)
    if u.shape[-1] < 6:
        raise IndexError("index 5 is out of bounds: encode_133171 needs >= 6 bits")
    out = np.empty(u.shape[:-1] + (2 * u.shape[-1],), dtype=np.int32)
    out[..., 0::2] = _conv_parity(u, _TAPS_133)
    out[..., 1::2] = _conv_parity(u, _TAPS_171)
    return out


def encode_57_batch(bits: np.ndarray) -> np.ndarray:
    """Vectorised twin of :func:`encode_57` (quirks included)."""
    u = np.asarray(bits)
    if not np.issubdtype(u.dtype, np.integer) or u.dtype.itemsize < 8:
        u = u.astype(np.int64)                    # the two start-up outputs below are sums, not parities
    if u.shape[-1] < 2:
        raise IndexError("index 1 is out of bounds: encode_57 needs >= 2 bits")
    out = np.empty(u.shape[:-1] + (2 * u.shape[-1],), dtype=np.int32)
    out[..., 0::2] = _conv_parity(u, (0, 2))
    out[..., 1::2] = _conv_parity(u, (0, 1, 2))
    out[..., 0] = u[..., 0]                       # no mod-2 on the very first output
    out[..., 1] = (u[..., 0] + u[..., 1]) % 2     # uses the NEXT bit
    return out


def encode_57(info_seq: np.ndarray) -> np.ndarray:
    """(7,5)-labelled K=3 rate-1/2 encoder of the reference
    (src/fnn_viterbi.py:122-154), bit-for-bit including its start-up quirk."""
    return encode_57_batch(np.asarray(info_seq).reshape(-1))


def encode_133171(bits: np.ndarray) -> np.ndarray:
    """(133,171) K=7 rate-1/2 encoder (src/fnn_viterbi.py:157-217)."""
    return encode_133171_batch(np.asarray(bits).reshape(-1))


def _bpsk(encoded: np.ndarray) -> np.ndarray:
    sym = np.asarray(encoded).astype(np.float32)
    sym[sym == 0] = -1
    return sym


def modulate_awgn(encoded_seq: np.ndarray, sigma: float) -> np.ndarray:
    """BPSK (0 -> -1, 1 -> +1) plus N(0, sigma^2) from the global numpy RNG;
    float64 result, one ``np.random.normal`` call (src/fnn_viterbi.py:224-253)."""
    sym = _bpsk(encoded_seq)
    return sym + np.random.normal(0, sigma, len(sym))


# ==============================================================================
# dataset  (reference: src/fnn_viterbi.py:260-406)
# ==============================================================================


def generate_class_bits(block_length: int) -> np.ndarray:
    """[2^b, b] uint8, row i = MSB-first binary digits of i
    (src/fnn_viterbi.py:260-287)."""
    idx = np.arange(2 ** block_length, dtype=np.int64)
    shifts = np.arange(block_length - 1, -1, -1, dtype=np.int64)
    return ((idx[:, None] >> shifts[None, :]) & 1).astype(np.uint8)


def create_one_hot_labels(labels: np.ndarray, num_classes: int) -> np.ndarray:
    """float32 one-hot [len(labels), num_classes] (src/fnn_viterbi.py:290-316)."""
    labels = np.asarray(labels)
    one_hot = np.zeros((len(labels), num_classes), dtype=np.float32)
    one_hot[np.arange(len(labels)), labels.astype(np.int64)] = 1.0
    return one_hot


def _noisy_class_major(codebook: np.ndarray, per_class: int, sigma) -> np.ndarray:
    """Every codeword repeated ``per_class`` times (class-major), BPSK + AWGN.

    The reference draws ``np.random.normal(0, sigma, 2b)`` once per row, rows in
    order (src/fnn_viterbi.py:369-388).  numpy's legacy generator produces the
    same stream for one [rows, 2b] draw, so the values are identical."""
    sym = np.repeat(_bpsk(codebook), per_class, axis=0)
    noise = np.random.normal(0, sigma, sym.shape)
    return (sym + noise).astype(np.float32)


def train_test_dataset(config: DecoderConfig) -> Tuple[np.ndarray, np.ndarray, np.ndarray,
                                                       np.ndarray, np.ndarray]:
    """(training_set, one-hot training labels, test_set, one-hot test labels,
    class_bits) exactly as the reference assembles them
    (src/fnn_viterbi.py:319-406): class-major rows, training rows drawn before
    test rows, float32 sets, float32 one-hot labels, (133,171) codebook."""
    class_bits = generate_class_bits(config.block_length)
    codebook = encode_133171_batch(class_bits)
    training_set = _noisy_class_major(codebook, config.samples_per_class_train, config.sigma)
    test_set = _noisy_class_major(codebook, config.samples_per_class_test, config.sigma)
    classes = np.arange(len(class_bits), dtype=np.int32)
    training_labels = np.repeat(classes, config.samples_per_class_train)
    test_labels = np.repeat(classes, config.samples_per_class_test)
    return (training_set, create_one_hot_labels(training_labels, config.n_classes),
            test_set, create_one_hot_labels(test_labels, config.n_classes), class_bits)


# ==============================================================================
# model  (reference: src/fnn_viterbi.py:413-510)
# ==============================================================================


def truncated_normal(shape, stddev: float, rng=None) -> np.ndarray:
    """tf.truncated_normal semantics: N(0, stddev^2) with values beyond two
    standard deviations re-drawn (src/fnn_viterbi.py:432-435)."""
    rng = np.random if rng is None else rng
    vals = rng.normal(0.0, 1.0, size=shape)
    bad = np.abs(vals) > 2.0
    while bad.any():
        vals[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(vals) > 2.0
    return (vals * stddev).astype(np.float32)


def layer_dims(config) -> list:
    """[n_input, hidden..., n_classes].  ``config.n_layers`` (optional, 1..3)
    keeps the first k hidden widths; the reference always uses three."""
    hidden = [config.n_hidden_1, config.n_hidden_2, config.n_hidden_3]
    k = int(getattr(config, "n_layers", 3))
    if not 1 <= k <= 3:
        raise ValueError("n_layers must be 1, 2 or 3")
    return [config.n_input] + hidden[:k] + [config.n_classes]


_W_KEYS = ("h1", "h2", "h3")
_B_KEYS = ("b1", "b2", "b3")


def initialize_weights(config, rng=None) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Weights ``{'h1','h2','h3','out'}`` as [in, out] float32 truncated normal
    with stddev sqrt(2/fan_in) and zero biases ``{'b1','b2','b3','out'}``
    (src/fnn_viterbi.py:413-468).  numpy arrays instead of TF variables; drawn
    from ``rng`` (default: the global numpy RNG)."""
    dims = layer_dims(config)
    weights, biases = {}, {}
    nh = len(dims) - 2
    for i in range(nh):
        weights[_W_KEYS[i]] = truncated_normal((dims[i], dims[i + 1]), math.sqrt(2.0 / dims[i]), rng)
        biases[_B_KEYS[i]] = np.zeros(dims[i + 1], dtype=np.float32)
    weights["out"] = truncated_normal((dims[-2], dims[-1]), math.sqrt(2.0 / dims[-2]), rng)
    biases["out"] = np.zeros(dims[-1], dtype=np.float32)
    return weights, biases


def weights_as_lists(weights: Dict, biases: Dict):
    """Dictionaries keyed like the reference's -> layer-ordered lists."""
    Ws = [weights[k] for k in _W_KEYS if k in weights] + [weights["out"]]
    bs = [biases[k] for k in _B_KEYS if k in biases] + [biases["out"]]
    return Ws, bs


def _to_device_input(x):
    import torch
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))
    return t.to(device="cuda", dtype=torch.float32).contiguous()


def _decoder_for(Ws, bs):
    """(packed network, is_wide).  The register-chained kernel when it has an instantiation for this network (the
    reference's hidden widths), else the wide HIP path, which takes any widths.  Both run on the GPU; there is no
    CPU path.  The packed network is kept by ``decoder.cached_network`` for as long as calls keep coming with the same
    weights (their bytes are the key): an SNR loop over ``predict`` / ``run_graph`` packs ONCE, where round 4 packed,
    uploaded and freed around every 0.4 ms decode.  Never closed here."""
    import torch
    from . import decoder
    decoder._require_cuda()              # no CPU path: fails loudly (NNViterbiError) without an MI355X
    net = decoder.cached_network(
        "ff", list(Ws) + list(bs),
        lambda: decoder._pack_any(Ws, bs, torch.device("cuda", torch.cuda.current_device()),
                                  decoder._resolve_precision(None)))
    return net, isinstance(net, decoder.WideMLP)


def multilayer_perceptron(x, weights: Dict, biases: Dict) -> np.ndarray:
    """tanh(...tanh(x@h1+b1)...@out+bout): the tanh'd class layer [N, n_classes]
    (src/fnn_viterbi.py:471-510), computed by the HIP kernel's debug output
    path.  Returns a numpy array (the reference returns a graph tensor)."""
    Ws, bs = weights_as_lists(weights, biases)
    net, wide = _decoder_for(Ws, bs)
    xd = _to_device_input(x)
    out = net.decode(xd, want_act=True)["act"] if wide else net.forward(xd)
    return out.cpu().numpy()


def predict(test_set, weights: Dict, biases: Dict, labels: Optional[np.ndarray] = None,
            tile_bits: Optional[int] = None):
    """``tf.argmax(pred, 1).eval({x: test_set})`` (src/fnn_viterbi.py:606-610)
    on the GPU.  Returns (predictions int64 [N], counters int64[4] or None)."""
    import torch
    Ws, bs = weights_as_lists(weights, biases)
    net, wide = _decoder_for(Ws, bs)
    if wide:   # hidden widths outside the fused kernel's instantiations: the wide path
        cls = net.decode(_to_device_input(test_set), want_cls64=True)["cls64"].cpu().numpy()
        counters = None
        if labels is not None:
            b = int(tile_bits) if tile_bits is not None else max(1, int(np.ceil(np.log2(Ws[-1].shape[1]))))
            diff = cls ^ np.asarray(labels).astype(np.int64)
            counters = np.array([int(_popcount(diff).sum()), int((diff != 0).sum()),
                                 len(cls) * b, len(cls)], dtype=np.int64)
        return cls, counters
    if not isinstance(test_set, torch.Tensor):
        # host arrays (the reference's call shape): pipelined PCIe transfers + decode
        return net.decode_host(np.asarray(test_set), labels, tile_bits=tile_bits)
    xd = _to_device_input(test_set)
    ld = None
    if labels is not None:
        ld = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).cuda()
    r = net.decode(xd, labels=ld, tile_bits=tile_bits, want_cls32=False, want_cls64=True)
    counters = None if r["counters"] is None else r["counters"].cpu().numpy()
    return r["cls64"].cpu().numpy(), counters


# ==============================================================================
# train + evaluate  (reference: src/fnn_viterbi.py:517-674)
# ==============================================================================


def run_network(config: DecoderConfig, training_set: np.ndarray, training_labels: np.ndarray,
                test_set: np.ndarray, test_labels: np.ndarray) -> np.ndarray:
    """Train, then decode the whole test set in one pass and print the accuracy
    (src/fnn_viterbi.py:517-616).  Returns int64 predictions."""
    from .training import train_network
    weights, biases = initialize_weights(config)
    weights, biases = train_network(config, training_set, training_labels, weights, biases)
    print("Optimization Finished!")
    labels_idx = np.argmax(test_labels, axis=1).astype(np.int32)
    predictions, counters = predict(test_set, weights, biases, labels=labels_idx,
                                    tile_bits=config.block_length)
    accuracy = (int(counters[3]) - int(counters[1])) / max(int(counters[3]), 1)
    print(f"Accuracy: {accuracy:.4f}")
    run_network.last_model = (weights, biases)
    return predictions


def _popcount(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint64)
    n = np.zeros(v.shape, dtype=np.int64)
    while v.any():
        n += (v & np.uint64(1)).astype(np.int64)
        v >>= np.uint64(1)
    return n


def calculate_error(test_labels: np.ndarray, predictions: np.ndarray, class_bits: np.ndarray,
                    config: DecoderConfig) -> None:
    """Prints accuracy, FER, BER and the bit-error count with the reference's
    exact format (src/fnn_viterbi.py:619-674).  Like the reference, the BER
    part assumes the class-major test order of ``train_test_dataset``: row r is
    compared with ``class_bits[r // samples_per_class_test]``."""
    predictions = np.asarray(predictions)
    test_labels_idx = np.argmax(test_labels, axis=1).astype(int)
    frame_errors = np.sum(predictions != test_labels_idx)
    fer = frame_errors / len(predictions)
    print(f"Accuracy: {1.0 - fer:.4f}")
    print(f"Frame Error Rate: {fer:.6f}")
    b = config.block_length
    n_used = config.num_classes * config.samples_per_class_test
    weights_msb = (1 << np.arange(b - 1, -1, -1)).astype(np.int64)
    true_val = np.asarray(class_bits[:config.num_classes]).astype(np.int64) @ weights_msb
    true_rows = np.repeat(true_val, config.samples_per_class_test)
    pred_rows = predictions[:n_used].astype(np.int64) & ((1 << b) - 1)
    bit_errors = int(_popcount(pred_rows ^ true_rows).sum())
    total_bits = config.num_test_samples * b
    ber = bit_errors / total_bits
    print(f"Bit Error Rate: {ber:.6f}")
    print(f"Total Bit Errors: {bit_errors}/{total_bits}")


def main() -> None:
    """Script entry of the reference (src/fnn_viterbi.py:681-749)."""
    print("=" * 70)
    print("Neural Network-Based Viterbi Decoder")
    print("=" * 70)
    config = DecoderConfig()
    print("\nConfiguration:")
    print(f"  Block Length: {config.block_length}")
    print(f"  SNR: {config.snr_db} dB")
    print(f"  Number of Classes: {config.num_classes}")
    print(f"  Training Samples: {config.num_training_samples}")
    print(f"  Test Samples: {config.num_test_samples}")
    print(f"  Network: {config.n_input} -> {config.n_hidden_1} -> "
          f"{config.n_hidden_2} -> {config.n_hidden_3} -> {config.n_classes}")
    print("\nGenerating datasets...")
    training_set, training_labels, test_set, test_labels, class_bits = train_test_dataset(config)
    print(f"Training set shape: {training_set.shape}")
    print(f"Test set shape: {test_set.shape}")
    print("\nTraining neural network...")
    predictions = run_network(config, training_set, training_labels, test_set, test_labels)
    print("\nEvaluating performance...")
    calculate_error(test_labels, predictions, class_bits, config)
    print("\n" + "=" * 70)
    print("Decoding complete!")
    print("=" * 70)


if __name__ == "__main__":
    main()
