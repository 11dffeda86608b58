"""GPU parity at the FULL sizes of BASELINE.json's configs 3 and 4 (config 2 is test_gpu_parity.py::
test_device_resident_batch_full_size, config 5 is the sharded path of test_distributed_cpu.py + bench.py --gpus 8).

The oracle cannot run these sizes in seconds, so each test pins the whole result through a property that does not
depend on the size, and a sample of the per-item results directly against the oracle."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from kzg_rs_amd import api
from kzg_rs_amd.api import KzgSettings

pytestmark = pytest.mark.gpu
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def settings():
    return KzgSettings.load_trusted_setup_file()


def test_config3_evaluate_16384_blobs(settings):
    """BASELINE config 3: evaluate_polynomial_in_evaluation_form (src/kzg_proof.rs:94-133) for 16384 device-resident
    blobs (2 GiB) in one call.
      - 48 sampled (blob, z) pairs, first and last included, against the oracle;
      - every 64th z is a root of unity: the result must be the blob's own element (the :104-108 early return), checked
        for all 256 of them from the blob bytes alone;
      - blobs [8192, 16384) repeat blobs [0, 8192) with the same z: both halves must agree bit for bit."""
    import torch
    n, half = 16384, 8192
    osettings = O.Settings.mainnet()
    g = torch.Generator(device="cuda").manual_seed(3)
    d_blobs = torch.empty((n, 131072), dtype=torch.uint8, device="cuda")
    d_blobs[:half] = torch.randint(0, 256, (half, 131072), dtype=torch.uint8, device="cuda", generator=g)
    d_blobs[:half, 0::32] &= 0x3F  # every element < 2^254 < r (canonical)
    d_blobs[half:] = d_blobs[:half]
    rng = random.Random(33)
    zs = [rng.randrange(R) for _ in range(half)]
    root_idx = {}
    for i in range(0, half, 64):
        k = rng.randrange(4096)
        root_idx[i] = k
        zs[i] = int.from_bytes(osettings.root(k), "big")
    z_le = np.frombuffer(b"".join(z.to_bytes(32, "little") for z in zs + zs), dtype=np.uint8).copy()
    d_z = torch.from_numpy(z_le).cuda()
    d_y = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    ms = settings.last_timings()[4]
    print("config 3: k_blob_evaluate over %d blobs: %.3f ms = %.0f blobs/s, %.0f GB/s of blob bytes" %
          (n, ms, n / ms * 1e3, n * 131072 / ms / 1e6))
    y = d_y.cpu().numpy().reshape(n, 32)
    assert (y[:half] == y[half:]).all()
    for i, k in root_idx.items():
        elem = d_blobs[i, 32 * k: 32 * k + 32].cpu().numpy().tobytes()
        assert y[i].tobytes()[::-1] == elem, (i, k)
    sample = [0, 1, 63, 65, half - 1, half + 7, n - 1] + [rng.randrange(n) for _ in range(41)]
    for i in sample:
        blob = d_blobs[i].cpu().numpy().tobytes()
        want = O.evaluate_polynomial_in_evaluation_form(blob, zs[i % half].to_bytes(32, "big"), osettings)
        assert y[i].tobytes()[::-1] == want, i


def test_config3_rejects_one_bad_element_among_16384(settings):
    """One non-canonical field element (== r) in the last blob of a 4096-blob call -> Err(BadArgs), src/dtypes.rs:48-57."""
    import torch
    n = 4096
    d_blobs = torch.zeros((n, 131072), dtype=torch.uint8, device="cuda")
    d_z = torch.ones(n * 32, dtype=torch.uint8, device="cuda")
    d_z.view(n, 32)[:, 31] = 0
    d_y = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    assert not d_y.any()  # the zero polynomial
    d_blobs[n - 1, 32 * 4095: 32 * 4096] = torch.tensor(list(R.to_bytes(32, "big")), dtype=torch.uint8)
    torch.cuda.synchronize()
    with pytest.raises(api.KzgError) as e:
        api.evaluate_polynomials_device(d_y.data_ptr(), d_blobs.data_ptr(), d_z.data_ptr(), n, settings)
    assert e.value.kind == "BadArgs"


def test_config4_msm_2_pow_20_points(settings):
    """BASELINE config 4: msm_variable_base (src/kzg_proof.rs:419,429,430) over 2^20 (point, scalar) pairs - the 4096
    G1 Lagrange points of the mainnet setup tiled 256 times, uniform random scalars.  By bilinearity the result must
    equal the 4096-term MSM over the distinct points with the 256 scalars of each point summed mod r, which the
    oracle computes."""
    ts = open(os.path.join(O.ROOT, "kzg_rs_amd", "data", "trusted_setup.txt")).read().split("\n")
    base = [bytes.fromhex(ts[2 + i]) for i in range(4096)]
    reps = 256
    n = 4096 * reps
    rng = np.random.Generator(np.random.PCG64(4))
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 0] &= 0x7F  # some scalars land in [r, 2^255): the entry point reduces them mod r like Scalar::from_raw
    got = api.g1_msm(base * reps, [sc[i].tobytes() for i in range(n)], settings)
    print("config 4: MSM kernels over 2^20 terms: %.1f ms" % settings.last_timings()[2])
    sums = [0] * 4096
    for i in range(n):
        sums[i & 4095] += int.from_bytes(sc[i].tobytes(), "big")
    want = O.g1_msm(b"".join(base), b"".join((s % R).to_bytes(32, "big") for s in sums), 4096)
    assert got == want
