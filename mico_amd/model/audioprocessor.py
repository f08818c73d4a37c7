"""Audio preprocessing with the reference's surface (model/audioprocessor.py:15-83).  Waveform loading, resampling and the Kaldi
log-mel filterbank are torchaudio code (absent from this image, unpinned third-party arithmetic) and stay on the host; everything
after the filterbank - normalise by (mean, 2 std), zero-pad to whole windows, cut into target_length windows, pick sample_num of
them - runs as one device kernel (mico_fbank_windows) on the [T, mel] filterbank."""
import os

import torch

from .. import _lib
from .videoprocessor import sample_indices, split  # noqa: F401  (same helper as audioprocessor.py:8-12)


class AudioProcessor(object):
    def __init__(self, melbins, target_length, sample_num, frame_shift=10, resize_melbin_num=224, mean=15.41663, std=6.55582,
                 training=True, device="cuda"):
        self.melbins = melbins
        self.target_length = target_length
        self.training = training
        self.frame_shift = frame_shift
        self.sample_num = sample_num
        self.resize_melbin_num = resize_melbin_num
        self.mean = mean
        self.std = std
        self.device = device

    def window_indices(self, src_length):
        """audioprocessor.py:50-62: number of target_length windows after padding, split into sample_num groups, one pick each."""
        pad_len = max(self.target_length * self.sample_num - src_length, self.target_length - src_length % self.target_length)
        total = (src_length + pad_len) // self.target_length
        return sample_indices(split(list(range(total)), self.sample_num), self.training)

    def from_fbank(self, fbank):
        """fbank [T, mel] (already resize_melbin_num wide) -> [sample_num, target_length, mel] on the device."""
        fb = fbank.to(self.device, torch.float32).contiguous()
        T, mel = fb.shape
        idx = torch.tensor(self.window_indices(T), dtype=torch.int32, device=fb.device)
        out = torch.empty((idx.numel(), self.target_length, mel), dtype=torch.float32, device=fb.device)
        rc = _lib.lib().mico_fbank_windows(fb.data_ptr(), T, mel, idx.data_ptr(), idx.numel(), self.target_length, float(self.mean),
                                           1.0 / (float(self.std) * 2), out.data_ptr(), torch.cuda.current_stream(fb.device).cuda_stream)
        _lib.check(rc, "mico_fbank_windows")
        return out

    def _host_fbank(self, torchaudio, wav_file):
        """16 kHz waveform scaled to int16 range -> Kaldi log-mel filterbank (25 ms frames, 10 ms shift), optionally resized along
        the mel axis (audioprocessor.py:34-43); third-party arithmetic, kept on the host."""
        wave, rate = torchaudio.load(wav_file)
        wave = wave if rate == 16000 else torchaudio.transforms.Resample(rate, 16000)(wave)
        fb = torchaudio.compliance.kaldi.fbank(wave * 2 ** 15, num_mel_bins=self.melbins, sample_frequency=16000, frame_length=25,
                                               frame_shift=10)
        if fb.size(1) != self.resize_melbin_num:
            fb = torch.nn.functional.interpolate(fb[None, None], size=(fb.size(0), self.resize_melbin_num), mode="bilinear")[0, 0]
        return fb

    def __call__(self, wav_file):
        if not os.path.exists(wav_file):
            print("not have audios", wav_file)
            return torch.zeros(self.sample_num, self.target_length, self.melbins)
        try:
            import torchaudio
        except ImportError as e:
            raise ImportError("AudioProcessor.__call__ needs torchaudio for decoding and the Kaldi filterbank; pass a filterbank "
                              "to from_fbank() instead") from e
        try:
            return self.from_fbank(self._host_fbank(torchaudio, wav_file))
        except Exception as e:   # audioprocessor.py:74-76
            print(e)
            return
