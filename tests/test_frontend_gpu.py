"""GPU: the log-mel front end (frontend.LogMelFrontEnd: reflect pad, STFT as an exact-fp32 MFMA GEMM over overlapping
frames, power/mel/dB, 87.5 %-overlap segmentation) against the oracle restatement of the reference's torchaudio pipeline."""
import math

import pytest
import torch

from oracle import ref_frontend

pytestmark = pytest.mark.gpu
CFG = {"fs": 16000, "n_fft": 1024, "win_len": 1024, "hop_len": 512, "n_mels": 64, "n_frames": 128, "overlap": 0.875}


def wave(seconds, seed):
    g = torch.Generator().manual_seed(seed)
    n = int(seconds * CFG["fs"])
    t = torch.arange(n) / CFG["fs"]
    x = 0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.sin(2 * math.pi * 3100.0 * t * (1 + 0.05 * t))
    x = x + 0.05 * torch.randn(n, generator=g)
    return (x * torch.hann_window(n, periodic=False).clamp_min(0.05)).float()     # loud middle, quiet edges


@pytest.mark.parametrize("seconds,seed", [(5.0, 0), (8.7, 1), (4.1, 2)])
def test_logmel_and_segments_match_oracle(seconds, seed):
    from neuralsampleid_amd.frontend import LogMelFrontEnd
    w = wave(seconds, seed)
    front = LogMelFrontEnd(CFG)
    lm = front.logmel(w.cuda()).cpu()
    ref = ref_frontend.logmel(w, CFG)
    assert lm.shape == ref.shape == (64, 1 + w.numel() // 512)
    # fp32 DFT-by-GEMM vs torch's FFT: agreement in dB; the tolerance is on the dB value, checked on every bin
    assert float((lm - ref).abs().max()) < 2e-3, float((lm - ref).abs().max())
    segs = front(w.cuda()).cpu()
    rsegs = ref_frontend.segments(w, CFG)
    assert segs.shape == rsegs.shape and segs.shape[1:] == (64, 128)
    assert float((segs - rsegs).abs().max()) < 2e-3
    if segs.shape[0] > 1:
        assert torch.equal(segs[1, :, :112], segs[0, :, 16:])                     # hop of 16 frames between segments


def test_silence_and_short_audio():
    from neuralsampleid_amd.frontend import LogMelFrontEnd
    front = LogMelFrontEnd(CFG)
    lm = front.logmel(torch.zeros(40000, device="cuda"))
    assert float((lm + 100.0).abs().max()) < 1e-4                                 # clamp(1e-10) -> -100 dB
    assert front(torch.zeros(20000, device="cuda")).shape == (0, 64, 128)         # 40 frames < one segment
    with pytest.raises(RuntimeError):
        front.logmel(torch.zeros(1000))                                           # host tensor


def test_mel_filterbank_restates_the_oracle():
    from neuralsampleid_amd.frontend import mel_filterbank
    a = mel_filterbank(513, 0.0, 8000.0, 64, 16000)
    b = ref_frontend.melscale_fbanks(513, 0.0, 8000.0, 64, 16000)
    assert torch.equal(a, b) and a.shape == (513, 64) and float(a.max()) <= 1.0 and bool((a.sum(0) > 0).all())
