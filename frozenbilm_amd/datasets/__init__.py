from .videotext_dataset import VideoText_Dataset, build_videotext_dataset, videotext_collate_fn  # noqa: F401
