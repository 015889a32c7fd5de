"""The demo.cu proving loop through the oracle on a tiny MLP: transcript sizes, quantised forward vs
float forward (SURVEY.md App. E 'Quantised forward'), and determinism."""
import numpy as np

import bigint_model as bm
from mlp_util import prove_mlp_oracle, synth_mlp


def test_tiny_mlp_transcript_and_forward(oracle):
    model = synth_mlp(oracle, [6, 5, 4], batch=3, seed=3)
    fr, g1 = prove_mlp_oracle(oracle, model)
    sh = model["shapes"]
    assert [s["n_gens"] for s in sh] == [8, 8] and [s["m"] for s in sh] == [8, 4]
    # FC: 3*k_in+2 + 1 + 1 ; ReLU (N = 4*8 = 32, k = 5): 3(k+5)+1 + 32 + 3(k+4)+1 + 16 + 3k+2
    assert fr.shape[0] == (3 * 3 + 4) + (31 + 32 + 28 + 16 + 17) + (3 * 3 + 4)
    assert g1.shape[0] == 2 * (1 + 3 * 3 + 1)
    # de-quantised Y_hat vs the float network: last layer is not rescaled => scale 2^32
    W0, W1, X = model["W"][0].astype(np.float64), model["W"][1].astype(np.float64), model["X"].astype(np.float64)
    want = np.maximum(X @ W0, 0) @ W1
    Y = model["_oracle"]["Yhat"].reshape(4, 4, 8)
    for b in range(3):
        for o in range(4):
            v = bm.limbs_to_int(Y[b, o])
            v = v - bm.R_MOD if v > bm.R_MOD // 2 else v
            assert abs(v / 2.0 ** 32 - want[b, o]) < 2e-3
    assert not Y[3].any()                       # padded batch row
    fr2, g12 = prove_mlp_oracle(oracle, model)
    assert np.array_equal(fr, fr2) and np.array_equal(g1, g12)
