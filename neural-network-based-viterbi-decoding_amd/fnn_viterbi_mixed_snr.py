"""Host-side mirror of the reference's ``src/fnn_viterbi_mixed_snr.py`` surface:
the mixed-SNR feed-forward decoder and the ``create_model_graph()`` /
``run_graph()`` inference surface the README names (README.md:31-40).

The forward pass + argmax + BER counting run in the HIP kernel
(``decoder.PackedMLP``); dataset assembly is vectorised numpy that reproduces
the reference's RNG draw order (golden vectors in tests/golden).

Reference behaviour kept on purpose:
  * ``binary_to_decimal`` is LSB-first (``2**arange``, :277-279) although
    ``generate_class_bits`` / ``extract_bits_single`` are MSB-first;
  * one-hot labels are float16 here (:371), float32 in ``fnn_viterbi``;
  * ``samples_per_snr = int(samples_per_class_train / n_snr)`` leaves the
    trailing rows of every class all-zero (:118, :430-444);
  * ``get_dataset`` shuffles the test set but does not return the shuffled
    labels (:492-513); they are exposed here as ``get_dataset.last_test_labels``;
  * the "random test set" is ONE long stream encoded once and chopped into
    2b-symbol rows (:464-475), so rows start from a non-zero encoder state.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import fnn_viterbi as _fv
from .fnn_viterbi import (encode_57, encode_133171, generate_class_bits,  # noqa: F401
                          generate_info_sequence, modulate_awgn, initialize_weights,
                          multilayer_perceptron, weights_as_lists)

# ==============================================================================
# configuration  (reference: src/fnn_viterbi_mixed_snr.py:33-123)
# ==============================================================================


class MixedSNRDecoderConfig:
    """Same keyword arguments and derived attributes as the reference's
    ``MixedSNRDecoderConfig`` (src/fnn_viterbi_mixed_snr.py:33-123)."""

    def __init__(self, block_length: int = 7, rate: int = 2, n_hidden_1: int = 128,
                 n_hidden_2: int = 64, n_hidden_3: int = 32, learning_rate: float = 0.0001,
                 training_epochs: int = 300, batch_size: int = 256,
                 beta_regularization: float = 0.001, samples_per_class_train: int = 1000,
                 samples_per_class_test: int = 10000, snr_db_range: Tuple[int, int] = (-2, 10),
                 test_snr_index: int = 0, use_dropout: bool = False,
                 dropout_keep_prob: float = 0.5, use_random_test_set: bool = False,
                 random_test_set_size: int = 5000):
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
        self.use_dropout = use_dropout
        self.dropout_keep_prob = dropout_keep_prob
        self.samples_per_class_train = samples_per_class_train
        self.samples_per_class_test = samples_per_class_test
        self.num_training_samples = self.num_classes * samples_per_class_train
        self.num_test_samples = self.num_classes * samples_per_class_test
        snr_db_vec = np.arange(snr_db_range[0], snr_db_range[1])
        self.snr_db_vec = snr_db_vec
        self.snr_vec = 10 ** (snr_db_vec / 10.0)
        self.n0_vec = 1.0 / self.snr_vec
        self.sigma_vec = np.sqrt(self.n0_vec * 0.5)
        self.test_snr_index = test_snr_index
        self.samples_per_snr = int(samples_per_class_train / len(snr_db_vec))
        self.use_random_test_set = use_random_test_set
        self.random_test_set_size = random_test_set_size
        self.random_test_length = block_length * random_test_set_size


# ==============================================================================
# utilities  (reference: src/fnn_viterbi_mixed_snr.py:263-328)
# ==============================================================================


def binary_to_decimal(binary_array: np.ndarray) -> int:
    """sum(bit[i] * 2**i): the FIRST element is the least significant bit
    (src/fnn_viterbi_mixed_snr.py:277-279)."""
    a = np.asarray(binary_array)
    return int(a.dot(2 ** np.arange(a.shape[0])))


def extract_bits_single(decimal_value: int, block_length: int) -> np.ndarray:
    """MSB-first uint8 digits of ``decimal_value`` padded to ``block_length``
    (src/fnn_viterbi_mixed_snr.py:297-300; a value needing more digits than
    ``block_length`` is an error there as well)."""
    digits = format(int(decimal_value), f"0{block_length}b")
    bits = np.zeros(block_length, dtype=np.uint8)
    bits[:len(digits)] = [int(ch) for ch in digits]
    return bits


def calculate_hamming_distance_decimal(val1: int, val2: int, block_length: int) -> int:
    """popcount(val1 XOR val2) (src/fnn_viterbi_mixed_snr.py:326-328;
    ``block_length`` is accepted and unused, as in the reference)."""
    return bin(int(val1) ^ int(val2)).count("1")


def create_one_hot_labels(labels: np.ndarray, num_classes: int) -> np.ndarray:
    """float16 one-hot (src/fnn_viterbi_mixed_snr.py:356-376)."""
    labels = np.asarray(labels)
    one_hot = np.zeros((len(labels), num_classes), dtype=np.float16)
    one_hot[np.arange(len(labels)), labels.astype(np.int64)] = 1.0
    return one_hot


# ==============================================================================
# dataset  (reference: src/fnn_viterbi_mixed_snr.py:379-513)
# ==============================================================================


def get_dataset(config: MixedSNRDecoderConfig):
    """(training_set, test_set, training one-hot, random-test one-hot,
    random_test_set, class_bits), same order, dtypes and RNG consumption as the
    reference (src/fnn_viterbi_mixed_snr.py:379-513)."""
    b = config.block_length
    class_bits = generate_class_bits(b)
    n_cls = len(class_bits)
    codebook = _fv.encode_133171_batch(class_bits)                    # [C, 2b] 0/1
    symbols = _fv._bpsk(codebook)                                     # [C, 2b] -1/+1 float32

    # training rows: class-major, then SNR, then sample; one N(0, sigma) draw
    # of 2b values per row in exactly that order (:430-444)
    n_snr, sps = len(config.snr_db_vec), config.samples_per_snr
    training_set = np.zeros((config.num_training_samples, 2 * b), dtype=np.float32)
    gauss = np.random.normal(0.0, 1.0, size=(n_cls, n_snr, sps, 2 * b))
    noisy = symbols[:, None, None, :] + config.sigma_vec[None, :, None, None] * gauss
    per_class = training_set.reshape(n_cls, config.samples_per_class_train, 2 * b)
    per_class[:, :n_snr * sps, :] = noisy.reshape(n_cls, n_snr * sps, 2 * b)
    training_labels = np.repeat(np.arange(n_cls, dtype=np.int32), config.samples_per_class_train)

    test_sigma = config.sigma_vec[config.test_snr_index]
    test_set = _fv._noisy_class_major(codebook, config.samples_per_class_test, test_sigma)
    test_labels = np.repeat(np.arange(n_cls, dtype=np.int32), config.samples_per_class_test)

    random_test_bits = np.random.randint(2, size=config.random_test_length)
    random_stream = modulate_awgn(encode_133171(random_test_bits), test_sigma)
    random_test_set = random_stream.reshape(config.random_test_set_size, 2 * b)
    rows = random_test_bits.reshape(config.random_test_set_size, b)
    lsb_weights = (2 ** np.arange(b)).astype(np.int64)
    random_labels = (rows.astype(np.int64) @ lsb_weights).astype(np.uint16)   # LSB-first (F8)

    order = np.random.permutation(len(test_labels))
    test_set = test_set[order, :]
    test_labels = test_labels[order]
    get_dataset.last_test_labels = test_labels

    return (training_set, test_set, create_one_hot_labels(training_labels, config.n_classes),
            create_one_hot_labels(random_labels, config.n_classes), random_test_set, class_bits)


# ==============================================================================
# metrics  (reference: src/fnn_viterbi_mixed_snr.py:634-673)
# ==============================================================================


def calculate_ber(predictions: np.ndarray, labels: np.ndarray, block_length: int) -> float:
    """sum popcount(pred XOR label) / (len(predictions) * block_length)
    (src/fnn_viterbi_mixed_snr.py:634-655)."""
    p = np.asarray(predictions).astype(np.int64)
    t = np.asarray(labels).astype(np.int64)
    n = min(len(p), len(t))                      # zip() stops at the shorter one
    errors = int(_fv._popcount(p[:n] ^ t[:n]).sum())
    return errors / (len(p) * block_length)


def calculate_accuracy(predictions: np.ndarray, labels: np.ndarray) -> float:
    """(predictions == labels).sum() / len (src/fnn_viterbi_mixed_snr.py:658-673)."""
    predictions = np.asarray(predictions)
    return np.sum(predictions == np.asarray(labels).astype(int)) / len(predictions)


# ==============================================================================
# model graph: create_model_graph() / run_graph()
# ==============================================================================

_VAR_NAMES = {"h1": "weights_h1", "h2": "weights_h2", "h3": "weights_h3", "out": "weights_out"}
_BIAS_NAMES = {"b1": "bias_h1", "b2": "bias_h2", "b3": "bias_h3", "out": "bias_out"}


def _ckpt_path(path: str) -> str:
    return path if path.endswith(".npz") else path + ".npz"


def save_model(path: str, weights: Dict, biases: Dict) -> str:
    """Checkpoint keyed by the reference's TF variable names
    (``weights_h1..out``, ``bias_h1..out``; src/fnn_viterbi.py:436-465)."""
    path = _ckpt_path(path)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    arrays = {_VAR_NAMES[k]: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
    arrays.update({_BIAS_NAMES[k]: np.asarray(v, dtype=np.float32) for k, v in biases.items()})
    np.savez(path, **arrays)
    return path


def load_model(path: str) -> Tuple[Dict, Dict]:
    data = np.load(_ckpt_path(path), allow_pickle=False)
    weights = {k: data[v] for k, v in _VAR_NAMES.items() if v in data.files}
    biases = {k: data[v] for k, v in _BIAS_NAMES.items() if v in data.files}
    return weights, biases


def multilayer_perceptron_with_dropout(x, weights: Dict, biases: Dict, keep_prob: float):
    """Training-time network with dropout after every hidden tanh
    (src/fnn_viterbi_mixed_snr.py:596-627).  Evaluation never uses it
    (:715-720); provided for API completeness as a PyTorch computation that
    returns a numpy array."""
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    Ws, bs = weights_as_lists(weights, biases)
    h = torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
    for W, b in zip(Ws[:-1], bs[:-1]):
        h = torch.tanh(h @ torch.as_tensor(W, device=dev) + torch.as_tensor(b, device=dev))
        h = torch.nn.functional.dropout(h, p=1.0 - float(keep_prob), training=True)
    out = torch.tanh(h @ torch.as_tensor(Ws[-1], device=dev) + torch.as_tensor(bs[-1], device=dev))
    return out.cpu().numpy()


class ModelGraph:
    """What ``create_model_graph`` returns and ``run_graph`` acts on."""

    def __init__(self, config, training_set, training_labels, test_set, test_labels,
                 random_test_set, random_test_labels, train_mode, model_save_path,
                 model_load_path):
        self.config = config
        self.training_set, self.training_labels = training_set, training_labels
        self.test_set, self.test_labels = test_set, test_labels
        self.random_test_set, self.random_test_labels = random_test_set, random_test_labels
        self.train_mode = train_mode
        self.model_save_path, self.model_load_path = model_save_path, model_load_path
        self.weights: Optional[Dict] = None
        self.biases: Optional[Dict] = None
        self.predictions: Optional[np.ndarray] = None
        self.accuracy: Optional[float] = None
        self.ber: Optional[float] = None
        self.has_run = False


def _evaluate(model: ModelGraph, x, one_hot_labels, title: str) -> None:
    cfg = model.config
    labels_idx = np.argmax(one_hot_labels, axis=1).astype(np.int32)
    preds, counters = _fv.predict(x, model.weights, model.biases, labels=labels_idx,
                                  tile_bits=cfg.block_length)
    model.predictions = preds
    model.accuracy = (int(counters[3]) - int(counters[1])) / max(int(counters[3]), 1)
    model.ber = int(counters[0]) / max(int(counters[2]), 1)
    print(f"{title} Accuracy: {model.accuracy:.4f}")
    print(f"{title} BER: {model.ber:.6f}")


def run_graph(model: ModelGraph) -> np.ndarray:
    """README.md:35-40.  ``train_mode=True``: train (once), optionally save,
    evaluate on the test set.  ``train_mode=False``: restore
    ``model_load_path`` and evaluate on the random or the standard test set
    (src/fnn_viterbi_mixed_snr.py:744-845).  Returns the predictions."""
    cfg = model.config
    if model.train_mode:
        if model.weights is None:
            from .training import train_network
            print("Starting training...")
            print(f"Training samples: {cfg.num_training_samples}")
            print(f"SNR values: {cfg.snr_db_vec} dB")
            print(f"Test SNR: {cfg.snr_db_vec[cfg.test_snr_index]} dB")
            w0, b0 = initialize_weights(cfg)
            model.weights, model.biases = train_network(
                cfg, model.training_set, model.training_labels, w0, b0,
                use_dropout=cfg.use_dropout, keep_prob=cfg.dropout_keep_prob)
            print("Optimization Finished!")
            if model.model_save_path:
                save_model(model.model_save_path, model.weights, model.biases)
                print(f"Model saved to {model.model_save_path}")
        print()
        _evaluate(model, model.test_set, model.test_labels, "Test Set")
    else:
        if model.weights is None:
            if not model.model_load_path:
                # the reference would evaluate uninitialised TF variables here and fail
                raise RuntimeError("train_mode=False needs model_load_path: "
                                   "there are no trained variables to evaluate")
            model.weights, model.biases = load_model(model.model_load_path)
            print(f"Model loaded from {model.model_load_path}")
        if cfg.use_random_test_set and model.random_test_set is not None:
            _evaluate(model, model.random_test_set, model.random_test_labels, "Random Test Set")
        else:
            _evaluate(model, model.test_set, model.test_labels, "Test Set")
    model.has_run = True
    return model.predictions


def create_model_graph(config: MixedSNRDecoderConfig, training_set: np.ndarray,
                       training_labels: np.ndarray, test_set: np.ndarray, test_labels: np.ndarray,
                       random_test_set: Optional[np.ndarray] = None,
                       random_test_labels: Optional[np.ndarray] = None, train_mode: bool = True,
                       model_save_path: Optional[str] = None,
                       model_load_path: Optional[str] = None, run: bool = True) -> ModelGraph:
    """Same positional/keyword signature as the reference
    (src/fnn_viterbi_mixed_snr.py:676-687).  The reference trains/evaluates
    inside this call and returns None; here the call does the same work (so
    code written against the reference behaves identically) AND returns the
    model handle that README's ``run_graph(model)`` acts on.  ``run=False``
    only builds the handle."""
    model = ModelGraph(config, training_set, training_labels, test_set, test_labels,
                       random_test_set, random_test_labels, train_mode, model_save_path,
                       model_load_path)
    if run:
        run_graph(model)
    return model


def main() -> None:
    """Script entry of the reference (src/fnn_viterbi_mixed_snr.py:852-930),
    except that the evaluation labels follow the shuffled test set."""
    print("=" * 70)
    print("Mixed SNR Neural Viterbi Decoder")
    print("=" * 70)
    config = MixedSNRDecoderConfig()
    print("\nGenerating datasets...")
    training_set, test_set, training_labels, random_test_labels, random_test_set, _ = \
        get_dataset(config)
    print(f"Training set shape: {training_set.shape}")
    print(f"Test set shape: {test_set.shape}")
    print("\nTraining neural network...")
    test_labels_onehot = create_one_hot_labels(get_dataset.last_test_labels, config.n_classes)
    create_model_graph(config=config, training_set=training_set, training_labels=training_labels,
                       test_set=test_set, test_labels=test_labels_onehot,
                       random_test_set=random_test_set, random_test_labels=random_test_labels,
                       train_mode=True, model_save_path="./models/mixed_snr_model")
    print("\n" + "=" * 70)
    print("Decoding complete!")
    print("=" * 70)


if __name__ == "__main__":
    main()
