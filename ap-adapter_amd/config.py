"""Task presets of the reference (/root/reference/config.py:1-82, ``get_config(task)``): which adapter scale, AudioMAE
pooling and guidance each editing task runs with, plus the prompts the reference driver loops over (inference.py:62-81).
Same keys and values, so a reference user's ``get_config(args.task)`` keeps working; the table form makes the
derived quantity the kernels care about explicit: ``La = (64 / time_pooling) * (8 / freq_pooling)`` audio tokens."""

# task: (ap_scale, time_pooling, freq_pooling, guidance_scale, positive prompts, negative prompt)
_PRESETS = {
    "timbre_transfer": (0.5, 2, 2, 7.5, ("a recording of a violin solo", "a recording of an acoustic guitar solo",
                                         "a recording of a harp solo"), "a recording of a piano solo"),
    "style_transfer": (0.55, 4, 4, 9.5, ("Jazz style music", "Rock style music", "Pop style music"), "Low quality"),
    "accompaniment_generation": (0.5, 2, 2, 7.5, ("Duet, Played with violin accompaniment", "Duet, Played with cello accompaniment",
                                                  "Duet, Played with flute accompaniment"), "solo"),
    "test": (0.5, 2, 2, 7.5, None, ""),
}
TASKS = tuple(_PRESETS)


def get_config(task):
    """dict with the reference's keys for ``task`` (config.py:2-82)"""
    if task not in _PRESETS:
        raise KeyError(f"unknown task {task!r}; one of {TASKS}")
    scale, tp, fp, gs, pos, neg = _PRESETS[task]
    return {"output_dir": task, "output_num_files": 1, "audio_prompt_file": "piano.wav", "ap_ckpt": "pytorch_model.bin",
            "ap_scale": scale, "time_pooling": tp, "freq_pooling": fp, "guidance_scale": gs,
            "positive_text_prompt": [""] if pos is None else [[t] for t in pos], "negative_text_prompt": [neg]}


def audio_tokens(cfg):
    """number of audio tokens the preset's pooling produces from AudioMAE's 64 x 8 patch grid (AudioMAE.py:148-182)"""
    return (64 // cfg["time_pooling"]) * (8 // cfg["freq_pooling"])
