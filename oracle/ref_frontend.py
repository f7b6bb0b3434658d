"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's evaluation front end (modules/transformations.py:27-34,
:94-105) — torchaudio.transforms.MelSpectrogram + AmplitudeToDB + transpose/unfold.

torchaudio (pinned 2.3.0 in the reference's requirements.txt:86) is ABSENT from this image, so the two torchaudio classes are
restated from their published algorithm: Spectrogram = torch.stft(n_fft, hop, win, hann_window(periodic), center=True,
pad_mode='reflect', normalized=False, onesided=True).abs()**2 — torch.stft itself IS available and is the very function
torchaudio calls, so the STFT half is pinned by the real implementation; MelScale = melscale_fbanks(htk, norm=None) as in
torchaudio/functional/functional.py; AmplitudeToDB(power) = 10*log10(clamp(x, 1e-10)) - 10*log10(max(1e-10, 1.0)).
PARITY OF THE MEL-FILTERBANK HALF IS PINNED ONLY BY THE PUBLISHED ALGORITHM (no torchaudio here to generate goldens)."""
import math

import torch


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))


def logmel(wave, cfg):
    """wave (L,) float32 CPU -> (n_mels, T) dB"""
    n_fft, hop, win = cfg["n_fft"], cfg["hop_len"], cfg["win_len"]
    spec = torch.stft(wave, n_fft, hop, win, torch.hann_window(win), center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True).abs().pow(2.0)                    # (n_freq, T)
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, float(cfg["fs"] // 2), cfg["n_mels"], cfg["fs"])
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)                        # MelScale.forward
    return 10.0 * torch.log10(torch.clamp(mel, min=1e-10)) - 10.0 * math.log10(max(1e-10, 1.0))


def segments(wave, cfg):
    """transformations.py:94-105 (train=False): (S, n_mels, n_frames)"""
    X = logmel(wave, cfg).transpose(1, 0)
    return X.unfold(0, size=cfg["n_frames"], step=int(cfg["n_frames"] * (1 - cfg["overlap"])))
