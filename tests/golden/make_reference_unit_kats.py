#!/usr/bin/env python3
"""Known answers for the calls the reference's OWN unit tests make (tests/test_encoders.py, tests/test_utilities.py),
obtained from the reference's shipped numpy functions.  Those test files cannot run (they import `*_improved` modules
that do not exist), and several of their expectations are false for the shipped code: this script
records what the shipped functions really return -- a value, or the exception type they raise -- for every call those
tests make with fixed arguments.  tests/test_reference_suite.py holds this package's helpers to the recorded answers.

Run ONLY in the build container (same import recipe as make_reference_golden.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_unit_kats.py

Output: tests/golden/reference_unit_kats.json (inputs and outputs only; no reference source text)."""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_reference_golden import _import_reference  # noqa: E402

# (module, function, positional args, keyword args); arrays are written as lists
CALLS = [
    ("fnn", "encode_57", [[1, 0, 1, 0, 1]], {}),
    ("fnn", "encode_57", [[0, 0, 0, 0, 0]], {}),
    ("fnn", "encode_57", [[1, 0, 0, 0, 0]], {}),
    ("fnn", "encode_133171", [[1, 0, 1, 0, 1, 1, 0]], {}),
    ("fnn", "encode_133171", [[0] * 10], {}),
    ("fnn", "encode_133171", [[1] * 10], {}),
    ("fnn", "encode_133171", [[1, 0, 1, 1, 0, 1, 0, 1, 1, 0]], {}),
    ("fnn", "encode_133171", [[1, 0, 1, 0, 1, 0, 1]], {}),
    ("fnn", "encode_133171", [[0, 1, 1, 0, 0, 1, 0]], {}),
    ("fnn", "encode_133171", [[1, 1, 0, 0, 1, 1, 1]], {}),
    ("fnn", "encode_133171", [[0, 1, 0, 1, 0, 1, 0]], {}),
    ("fnn", "encode_133171", [[0, 0, 0, 0, 0, 0, 1, 0, 1, 1, 0]], {}),
    ("fnn", "encode_133171", [[0] * 5], {}),                      # zero_input_zero_output, length 5: raises
    ("fnn", "encode_133171", [[1]], {}),                          # test_encode_single_bit: raises
    ("fnn", "encode_133171", [[0, 1] * 10], {}),
    ("bit", "extract_state_bits", [0], {"constraint_length": 6}),
    ("bit", "extract_state_bits", [63, 6], {}),
    ("bit", "extract_state_bits", [5], {"constraint_length": 6}),
    ("bit", "encode_133171_with_state", [[1, 0, 1, 1, 0]], {"state_num": 0}),
    ("bit", "encode_133171_with_state", [[1, 0, 1]], {"state_num": 0}),
    ("bit", "encode_133171_with_state", [[1, 0, 1]], {"state_num": 1}),
    ("bit", "encode_133171_with_state", [[1, 0, 1]], {"state_num": 100}),     # test_encode_with_invalid_state: raises
    ("fnn", "generate_class_bits", [3], {}),
    ("fnn", "generate_class_bits", [1], {}),
    ("fnn", "create_one_hot_labels", [[0, 1, 2]], {"num_classes": 3}),
    ("fnn", "create_one_hot_labels", [[1, 3, 5, 7]], {"num_classes": 10}),
    ("fnn", "create_one_hot_labels", [[3]], {"num_classes": 5}),
    ("mix", "binary_to_decimal", [[0, 0, 0]], {}),
    ("mix", "binary_to_decimal", [[0, 0, 1]], {}),
    ("mix", "binary_to_decimal", [[0, 1, 0]], {}),
    ("mix", "binary_to_decimal", [[1, 0, 0]], {}),
    ("mix", "binary_to_decimal", [[1] * 5], {}),
    ("mix", "binary_to_decimal", [[0] * 8], {}),
    ("mix", "binary_to_decimal", [[0]], {}),
    ("mix", "binary_to_decimal", [[1]], {}),
    ("mix", "binary_to_decimal", [[1, 0, 1, 0, 1, 0, 1, 0]], {}),
    ("mix", "extract_bits_single", [0], {"block_length": 5}),
    ("mix", "extract_bits_single", [5], {"block_length": 5}),
    ("mix", "extract_bits_single", [127, 7], {}),
    ("mix", "extract_bits_single", [1], {"block_length": 7}),
] + [("mix", "extract_bits_single", [v], {"block_length": 7}) for v in (0, 5, 15, 31, 63)] + [
    ("mix", "calculate_hamming_distance_decimal", [0, 1], {"block_length": 3}),
    ("mix", "calculate_hamming_distance_decimal", [0, 15], {"block_length": 4}),
    ("mix", "calculate_hamming_distance_decimal", [5, 3], {"block_length": 7}),
    ("mix", "calculate_hamming_distance_decimal", [3, 5], {"block_length": 7}),
    ("mix", "calculate_hamming_distance_decimal", [5, 3], {"block_length": 3}),
    ("mix", "calculate_hamming_distance_decimal", [0, 5, 4], {}),
    ("mix", "calculate_hamming_distance_decimal", [5, 7, 4], {}),
    ("mix", "calculate_hamming_distance_decimal", [0, 7, 4], {}),
    ("mix", "calculate_hamming_distance_decimal", [0, 0], {"block_length": 0}),
] + [("mix", "calculate_hamming_distance_decimal", [v, v], {"block_length": 7}) for v in (0, 5, 15, 31)]


def _arg(a):
    return np.array(a) if isinstance(a, list) else a


def main():
    ref_fnn, ref_mix, _, ref_bit = _import_reference()
    mods = {"fnn": ref_fnn, "mix": ref_mix, "bit": ref_bit}
    out = []
    for mod, fn, args, kwargs in CALLS:
        rec = {"module": mod, "function": fn, "args": args, "kwargs": kwargs}
        try:
            r = getattr(mods[mod], fn)(*[_arg(a) for a in args], **kwargs)
            if isinstance(r, np.ndarray):
                rec["result"] = r.tolist()
                rec["dtype"] = str(r.dtype)
            else:
                rec["result"] = int(r) if isinstance(r, (int, np.integer)) else float(r)
        except Exception as e:                      # noqa: BLE001  (the exception TYPE is the known answer)
            rec["raises"] = type(e).__name__
        out.append(rec)
    # the round trip the reference's tests assume (extract_bits_single then binary_to_decimal), all 7-bit values
    out.append({"module": "mix", "function": "round_trip_7", "args": [], "kwargs": {},
                "result": [int(ref_mix.binary_to_decimal(ref_mix.extract_bits_single(v, 7))) for v in range(128)]})
    with open(os.path.join(HERE, "reference_unit_kats.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote {len(out)} known answers; {sum('raises' in r for r in out)} of them exceptions")


if __name__ == "__main__":
    main()
