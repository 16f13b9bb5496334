"""Drop-in model classes: same import paths as the reference's ``stage2_accompaniment/model`` package
(``model.music_performer.MusicPerformer``, ``model.music_gpt2.MusicGPT2``)."""
