from .videotext_dataset import (PackedVideoText_Dataset, VideoText_Dataset, build_videotext_dataset,  # noqa: F401
                                packed_collate_fn, stage_packed_batch, videotext_collate_fn)
