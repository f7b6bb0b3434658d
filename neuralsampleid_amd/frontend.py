"""Log-mel front end on the MI355X: waveform -> (S, n_mels, n_frames) segments for the encoder.

Mirrors the evaluation branch of the reference's GPUTransformSampleID (modules/transformations.py:27-34 and :94-105):
    MelSpectrogram(sample_rate=fs, win_length, hop_length, n_fft, n_mels)   torchaudio 2.3.0 defaults: f_min 0, f_max fs/2,
        center=True / reflect pad, periodic Hann window, power 2.0, HTK mel scale, norm None, onesided
    AmplitudeToDB()                                                         stype power: 10*log10(clamp(x, 1e-10)), ref 1
    X.transpose -> unfold(0, size=n_frames, step=int(n_frames*(1-overlap))) -> (S, n_mels, n_frames)

MI355X form: the STFT is one exact-fp32 MFMA GEMM — frames are overlapping rows of the reflect-padded waveform (row stride
= hop, no frame matrix is materialised) against the constant matrix [hann*cos ; -hann*sin] — followed by a per-frame
power/mel/dB kernel and the segment gather (csrc/misc.hip). Everything is stream-ordered; no host sync."""
import math

import torch

from . import ops
from ._lib import call


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated: (n_freqs, n_mels), float32 arithmetic
    in the same order (triangles from the slopes between the mel-spaced centre frequencies)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class LogMelFrontEnd:
    """front = LogMelFrontEnd(cfg, device); segs = front(wave)  — wave (L,) fp32 on the GPU, segs (S, n_mels, n_frames)."""

    def __init__(self, cfg: dict, device="cuda"):
        self.fs, self.n_fft, self.hop = int(cfg["fs"]), int(cfg["n_fft"]), int(cfg["hop_len"])
        self.win = int(cfg.get("win_len", self.n_fft))
        self.n_mels, self.n_frames = int(cfg["n_mels"]), int(cfg["n_frames"])
        self.step = int(self.n_frames * (1 - float(cfg["overlap"])))          # transformations.py:102
        if self.win != self.n_fft:
            raise NotImplementedError("win_length != n_fft (the reference config uses 1024/1024)")
        if self.n_fft % 4 or self.hop % 4:
            raise NotImplementedError("n_fft and hop_len must be multiples of 4 (16-byte rows of the framing GEMM)")
        self.n_freq = self.n_fft // 2 + 1
        self.device = torch.device(device)
        # DFT matrix with the periodic Hann window folded in, built in fp64: rows [0, n_freq) = w*cos, then -w*sin
        n = torch.arange(self.n_fft, dtype=torch.float64)
        k = torch.arange(self.n_freq, dtype=torch.float64).unsqueeze(1)
        win = torch.hann_window(self.n_fft, periodic=True, dtype=torch.float64)
        ang = 2.0 * math.pi * k * n / self.n_fft
        rows = 2 * self.n_freq
        self.ld = (rows + 3) // 4 * 4
        W = torch.zeros(self.ld, self.n_fft, dtype=torch.float64)
        W[:self.n_freq] = win * torch.cos(ang)
        W[self.n_freq:rows] = -win * torch.sin(ang)
        self.W = W.to(torch.float32).to(self.device).contiguous()
        fb = mel_filterbank(self.n_freq, 0.0, float(self.fs // 2), self.n_mels, self.fs).t().contiguous()   # (n_mels, n_freq)
        nz = fb > 0
        lo = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(self.n_mels, dtype=torch.long))
        hi = torch.where(nz.any(1), self.n_freq - nz.flip(1).float().argmax(1), torch.zeros(self.n_mels, dtype=torch.long))
        self.fb = fb.to(self.device)
        self.band = torch.stack((lo, hi), 1).to(torch.int32).contiguous().to(self.device)

    def n_frames_of(self, L: int) -> int:
        return 1 + L // self.hop                                             # torch.stft, center=True

    def logmel(self, wave: torch.Tensor) -> torch.Tensor:
        """(L,) fp32 on the GPU -> (n_mels, T) dB"""
        if wave.dim() != 1 or not wave.is_cuda or wave.dtype != torch.float32:
            raise RuntimeError("the front end takes one mono fp32 waveform on the MI355X device")
        wave = wave.contiguous()
        L, pad = wave.numel(), self.n_fft // 2
        T = self.n_frames_of(L)
        s = ops._stream()
        padded = torch.empty(L + 2 * pad, device=wave.device, dtype=torch.float32)
        call("nsid_reflect_pad", ops._p(wave), L, pad, ops._p(padded), s)
        prec = ops.get_gemm_precision()
        ops.set_gemm_precision("fp32")                                       # dB dynamic range needs exact fp32 products
        try:
            spec = torch.empty((T, self.ld), device=wave.device, dtype=torch.float32)
            call("nsid_linear_fwd", ops._p(padded), self.hop, ops._p(self.W), ops.F32, None, ops._p(spec), self.ld, T,
                 self.ld, self.n_fft, 1, None, None, ops.ACT_NONE, ops.ACT_NONE, None, 1, ops.F32, s)
        finally:
            ops.set_gemm_precision(prec)
        out = torch.empty((self.n_mels, T), device=wave.device, dtype=torch.float32)
        call("nsid_power_mel_db", ops._p(spec), self.ld, self.n_freq, ops._p(self.fb), ops._p(self.band), self.n_mels, T,
             ops._p(out), s)
        return out

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        """(L,) -> (S, n_mels, n_frames); S = 0 rows when the audio is shorter than one segment (the reference's unfold
        raises there and its caller skips the file, transformations.py:101-104)"""
        lm = self.logmel(wave)
        T = lm.shape[1]
        S = (T - self.n_frames) // self.step + 1 if T >= self.n_frames else 0
        out = torch.empty((S, self.n_mels, self.n_frames), device=wave.device, dtype=torch.float32)
        if S > 0:
            call("nsid_unfold_segments", ops._p(lm), self.n_mels, T, self.n_frames, self.step, S, ops._p(out),
                 ops._stream())
        return out
