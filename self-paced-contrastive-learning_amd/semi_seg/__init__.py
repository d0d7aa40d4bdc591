"""Mirror package of the reference's ``semi_seg`` (hot-path modules only) + the per-data-set tables a driver reads from the
package itself (semi_seg/__init__.py:1-88: ``from semi_seg import ratio_zoo`` in main_pretrain_encoder.py:12) -- kept for
the two data sets the mirror's data path covers."""
# labelled scan counts the fine-tune stage is run with, per data set (semi_seg/__init__.py:6-11,32-38)
ratio_zoo = {"acdc": [1, 2, 4, 174], "prostate": [3, 5, 7, 40]}
pre_max_epoch_zoo = {"acdc": 80, "prostate": 80}
ft_max_epoch_zoo = {"acdc": 60, "prostate": 80}
num_batches_zoo = {"acdc": 200, "prostate": 300}
data2class_numbers = {"acdc": 4, "prostate": 2}
data2input_dim = {"acdc": 1, "prostate": 1}
pre_lr_zooms = {"acdc": 0.0000005, "prostate": 0.0000005}
ft_lr_zooms = {"acdc": 0.0000002, "prostate": 0.0000005}


def __getattr__(name):
    if name == "labeled_filenames":  # (lives with the splitter that reads it; imported lazily: the data package needs the GPU library)
        from .data.creator import labeled_filenames
        return labeled_filenames
    raise AttributeError(name)
