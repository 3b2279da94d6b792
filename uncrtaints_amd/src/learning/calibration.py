"""Calibration of the predicted variances with the reference's surface (model/train_reconstruct.py:475-530): `compute_ece`
(error of the most-certain fraction of the samples) and `compute_uce_auce` (binned |error - uncertainty|).

Host logic, like in the reference: the inputs are Python lists with ONE scalar per test sample (sample-averaged variance
and error, collected by the evaluation loop from `img_metrics`, train_reconstruct.py:300-330), so there is nothing for
the GPU to do here.  The reference's matplotlib / TensorBoard side effects are not part of this module."""
import numpy as np


def _f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32)      # torch.Tensor(list) stores float32


def compute_ece(vars, errors, n_samples, percent=5):
    """train_reconstruct.py:475-487: sort the samples by ascending uncertainty; entry i is the nan-mean error of the
    first (i+1)*percent % of them.  Returns a numpy array of 100//percent floats."""
    v, e = _f32(vars), _f32(errors)
    order = np.argsort(v, kind="stable")
    es = e[order].astype(np.float64)
    n_steps = 100 // percent
    # integer bin ends exactly as torch.linspace(0, n_samples, n_steps + 1, dtype=int)[1:] produces them
    step = np.float32(n_samples) / np.float32(n_steps)
    half = (n_steps + 1) // 2
    ends = [int(np.float32(i) * step) if i < half else int(np.float32(n_samples) - step * np.float32(n_steps - i))
            for i in range(1, n_steps + 1)]
    ok = ~np.isnan(es)
    csum = np.concatenate([[0.0], np.cumsum(np.where(ok, es, 0.0))])
    ccnt = np.concatenate([[0], np.cumsum(ok)])
    with np.errstate(invalid="ignore", divide="ignore"):
        return (csum[ends] / ccnt[ends]).astype(np.float32)


def compute_uce_auce(var, errors, n_samples, percent=5, l2=True, mode="val", step=0):
    """train_reconstruct.py:492-530: bin the samples into 100//percent equal-width uncertainty bins between the smallest
    and the largest variance; per bin compare sqrt(mean var) with the RMSE (l2) or mean std with the MAE (l1).
    UCE weighs the bins by their share of `n_samples`, AUCE is the plain mean over non-empty bins.  -> (uce, auce)"""
    n_bins = 100 // percent
    v, e = _f32(var), _f32(errors)
    edges = np.linspace(float(v.min()), float(v.max()), num=n_bins)[1:]
    idx = np.digitize(v, bins=edges)                                     # 0 .. n_bins-1
    cnt = np.bincount(idx, minlength=n_bins).astype(np.float64)
    sd = np.sqrt(v).astype(np.float64)
    ed = e.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        if l2:
            bk_var = np.sqrt(np.bincount(idx, weights=sd * sd, minlength=n_bins) / cnt)
            bk_err = np.sqrt(np.bincount(idx, weights=ed * ed, minlength=n_bins) / cnt)
        else:
            bk_var = np.bincount(idx, weights=np.abs(sd), minlength=n_bins) / cnt
            bk_err = np.bincount(idx, weights=np.abs(ed), minlength=n_bins) / cnt
    calib = np.abs(bk_err - bk_var)                                      # NaN for empty bins and bins holding a NaN error
    # the reference weighs with torch.histogram(bin index, n_bins) over [min index, max index]
    weight = np.histogram(idx.astype(np.float32), bins=n_bins)[0] / float(n_samples)
    uce = float(np.nansum(weight * calib))
    auce = float(np.nanmean(calib))
    return uce, auce
