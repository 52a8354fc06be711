"""Parameter sets of the WoPBS S-Box path.

``PARAM_OPT`` restates the reference's ``WopbsParameters`` constant
(/root/reference/src/client/client.rs:31-57): it fixes every shape on the hot path.
``PARAM_TOY`` keeps the decomposition (bases / levels) and N = 512 so the same kernels
run, but shrinks n and k and the noise so that the CPU oracle finishes a whole AES
block in seconds (the reference suggests exactly this at src/main.rs:74-75).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass


class CParams(ctypes.Structure):
    """``fheaes_params`` of include/fheaes.h (and ``orc_params`` of the oracle: same layout)."""

    _fields_ = [
        ("lwe_dimension", ctypes.c_uint32),
        ("glwe_dimension", ctypes.c_uint32),
        ("polynomial_size", ctypes.c_uint32),
        ("pbs_base_log", ctypes.c_uint32),
        ("pbs_level", ctypes.c_uint32),
        ("ks_base_log", ctypes.c_uint32),
        ("ks_level", ctypes.c_uint32),
        ("pfks_base_log", ctypes.c_uint32),
        ("pfks_level", ctypes.c_uint32),
        ("cbs_base_log", ctypes.c_uint32),
        ("cbs_level", ctypes.c_uint32),
    ]


@dataclass(frozen=True)
class WopbsParameters:
    name: str
    lwe_dimension: int
    glwe_dimension: int
    polynomial_size: int
    lwe_noise_std: float
    glwe_noise_std: float
    pbs_base_log: int
    pbs_level: int
    ks_level: int
    ks_base_log: int
    pfks_level: int
    pfks_base_log: int
    pfks_noise_std: float
    cbs_level: int
    cbs_base_log: int
    message_modulus: int = 2
    carry_modulus: int = 1

    # derived shapes ------------------------------------------------------
    @property
    def n(self) -> int:
        return self.lwe_dimension

    @property
    def k(self) -> int:
        return self.glwe_dimension

    @property
    def N(self) -> int:
        return self.polynomial_size

    @property
    def big(self) -> int:
        """dimension of the big LWE key (kN); ciphertexts of the API live under it."""
        return self.glwe_dimension * self.polynomial_size

    @property
    def big1(self) -> int:
        return self.big + 1

    @property
    def ksk_words(self) -> int:
        return self.big * self.ks_level * (self.n + 1)

    @property
    def bsk_words(self) -> int:
        return self.n * self.pbs_level * (self.k + 1) ** 2 * self.N

    @property
    def pfpksk_words(self) -> int:
        return (self.k + 1) * self.big1 * self.pfks_level * (self.k + 1) * self.N

    @property
    def ggsw_words(self) -> int:
        """one circuit-bootstrapped GGSW (standard domain), per input bit"""
        return self.cbs_level * (self.k + 1) ** 2 * self.N

    @property
    def key_bytes_per_bit(self) -> int:
        """key bytes streamed by one bit circuit-bootstrap without reuse"""
        return 8 * (self.ksk_words + self.bsk_words + self.pfpksk_words)

    def c_struct(self) -> CParams:
        return CParams(
            self.lwe_dimension, self.glwe_dimension, self.polynomial_size,
            self.pbs_base_log, self.pbs_level, self.ks_base_log, self.ks_level,
            self.pfks_base_log, self.pfks_level, self.cbs_base_log, self.cbs_level,
        )


# client.rs:31-57
PARAM_OPT = WopbsParameters(
    name="PARAM_OPT",
    lwe_dimension=669,
    glwe_dimension=4,
    polynomial_size=512,
    lwe_noise_std=3.0517578125e-05,
    glwe_noise_std=3.162026630747649e-16,
    pbs_base_log=8,
    pbs_level=5,
    ks_level=6,
    ks_base_log=2,
    pfks_level=3,
    pfks_base_log=12,
    pfks_noise_std=3.162026630747649e-16,
    cbs_level=1,
    cbs_base_log=15,
)

# Same gadget shapes, tiny dimensions and noise: NOT secure, test-only.
PARAM_TOY = WopbsParameters(
    name="PARAM_TOY",
    lwe_dimension=24,
    glwe_dimension=1,
    polynomial_size=512,
    lwe_noise_std=2.0 ** -30,
    glwe_noise_std=2.0 ** -50,
    pbs_base_log=8,
    pbs_level=5,
    ks_level=6,
    ks_base_log=2,
    pfks_level=3,
    pfks_base_log=12,
    pfks_noise_std=2.0 ** -50,
    cbs_level=1,
    cbs_base_log=15,
)
