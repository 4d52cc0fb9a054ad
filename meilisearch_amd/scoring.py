"""Bindings of the host-side scoring arithmetic of libmsi (a3, a14, a15)."""
import ctypes as C

import numpy as np

from ._lib import lib
from .device import np_ptr


def distribution_shift(mean, sigma, score):
    """DistributionShift::shift — crates/milli/src/vector/distribution.rs:103-130."""
    return float(np.float32(lib().msi_distribution_shift(mean, sigma, score)))


def rank_global_score(pairs):
    """Rank::global_score — crates/milli/src/score_details.rs:517-546."""
    r = np.array([p[0] for p in pairs], dtype=np.uint32)
    m = np.array([p[1] for p in pairs], dtype=np.uint32)
    return float(lib().msi_rank_global_score(np_ptr(r) if r.size else None,
                                             np_ptr(m) if m.size else None, len(pairs)))


def compare_scores(left, left_ratio, right, right_ratio):
    """compare_scores over Score sequences — crates/milli/src/search/hybrid.rs:32-80."""
    l = np.array(left, dtype=np.float64)
    r = np.array(right, dtype=np.float64)
    return int(lib().msi_compare_scores(np_ptr(l) if l.size else None, len(left), left_ratio,
                                        np_ptr(r) if r.size else None, len(right), right_ratio))
