"""Host logic of the preprocessing row (f3) against the reference's own code (tests/golden/processors.pt): `split` and the
evaluation-mode window choice of the audio processor."""
from common import golden


def test_split_and_window_choice():
    from mico_amd.model.audioprocessor import AudioProcessor
    from mico_amd.model.videoprocessor import sample_indices, split
    fx = golden("processors.pt")
    for (n, k), want in fx["split"].items():
        assert split(list(range(n)), k) == want
    assert sample_indices([[0, 1, 2], [3, 4], [5]], training=False) == [1, 3, 5]
    for case in fx["audio"]:
        proc = AudioProcessor(case["melbins"], case["target_length"], case["sample_num"], resize_melbin_num=case["melbins"],
                              training=False, device="cpu")
        idx = proc.window_indices(case["fbank"].shape[0])
        assert len(idx) == case["sample_num"] == case["out"].shape[0]
