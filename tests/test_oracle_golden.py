"""oracle/vvenc_oracle.c against the committed golden vectors (outputs of the reference itself,
tests/golden/*.npz written by tests/gen_golden.py).  CPU only; runs where /root/reference is absent."""
import golden_replay as G


def test_distortion(oracle):
    G.check_distortion(oracle)


def test_distortion_ext(oracle):
    G.check_distortion_ext(oracle)


def test_transform_matrices(oracle):
    G.check_transform_matrices(oracle)


def test_transform(oracle):
    G.check_transform(oracle)


def test_scan(oracle):
    G.check_scan(oracle)


def test_quant(oracle):
    G.check_quant(oracle)


def test_quant_qp(oracle):
    G.check_quant_qp(oracle)


def test_mctf_kernels(oracle):
    G.check_mctf_kernels(oracle)


def test_mctf_me(oracle):
    G.check_mctf_me(oracle)


def test_interp(oracle):
    G.check_interp(oracle)


def test_mctf_apply(oracle):
    G.check_mctf_apply(oracle)


def test_dmvr(oracle):
    G.check_dmvr(oracle)


def test_alf(oracle):
    G.check_alf(oracle)


def test_alf_filter(oracle):
    G.check_alf_filter(oracle)
