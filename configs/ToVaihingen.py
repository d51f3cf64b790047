"""Data-side config of the Potsdam -> Vaihingen task: the attribute surface of the reference's
configs/ToVaihingen.py:5-127 (names and values).  Paths point at the ISPRS tile folders the reference's
convert_datasets/ scripts produce; this build's bench/CLI feed synthetic tensors of the same contract
(regda_amd/synthetic.py), so the augmentation pipelines are described declaratively instead of through
albumentations objects."""
DATASETS = 'IsprsDA'
TARGET_SET = 'Vaihingen'
SRC_MEAN = (97.4603, 86.3828, 92.4078)       # source = Potsdam (ToVaihingen.py:51-52)
SRC_STD = (36.2062, 35.7308, 35.3348)
MEAN = (120.8217, 81.8250, 81.2344)          # target = Vaihingen (ToVaihingen.py:73-74)
STD = (54.7461, 39.3116, 37.9288)

source_dir = dict(image_dir=['data/IsprsDA/Potsdam/img_dir/train'], mask_dir=['data/IsprsDA/Potsdam/ann_dir/train'])
target_dir = dict(image_dir=['data/IsprsDA/Vaihingen/img_dir/train'], mask_dir=['data/IsprsDA/Vaihingen/ann_dir/train'])
val_dir = dict(image_dir=['data/IsprsDA/Vaihingen/img_dir/val'], mask_dir=['data/IsprsDA/Vaihingen/ann_dir/val'])
test_dir = dict(image_dir=['data/IsprsDA/Vaihingen/img_dir/test'], mask_dir=['data/IsprsDA/Vaihingen/ann_dir/test'])

_TRAIN_AUG = [('RandomCrop', (512, 512)), ('OneOf', ('HorizontalFlip', 'VerticalFlip', 'RandomRotate90'), 0.75),
              ('Normalize', dict(mean=SRC_MEAN, std=SRC_STD, max_pixel_value=1)), ('ToTensor',)]
_EVAL_AUG = [('Normalize', dict(mean=MEAN, std=STD, max_pixel_value=1)), ('ToTensor',)]

SOURCE_DATA_CONFIG = dict(image_dir=source_dir['image_dir'], mask_dir=source_dir['mask_dir'], transforms=_TRAIN_AUG,
                          CV=dict(k=10, i=-1), training=True, batch_size=8, num_workers=4)
_TARGET_AUG = [('RandomCrop', (512, 512)), ('OneOf', ('HorizontalFlip', 'VerticalFlip', 'RandomRotate90'), 0.75),
               ('Normalize', dict(mean=MEAN, std=STD, max_pixel_value=1)), ('ToTensor',)]
TARGET_DATA_CONFIG = dict(image_dir=target_dir['image_dir'], mask_dir=target_dir['mask_dir'], transforms=_TARGET_AUG,
                          CV=dict(k=10, i=-1), training=True, batch_size=8, num_workers=4)
PSEUDO_DATA_CONFIG = dict(image_dir=target_dir['image_dir'], mask_dir=target_dir['mask_dir'], transforms=_EVAL_AUG,
                          CV=dict(k=10, i=-1), training=False, batch_size=1, num_workers=1)
EVAL_DATA_CONFIG = dict(image_dir=val_dir['image_dir'], mask_dir=val_dir['mask_dir'], transforms=_EVAL_AUG,
                        CV=dict(k=10, i=-1), training=False, batch_size=1, num_workers=1)
TEST_DATA_CONFIG = dict(image_dir=test_dir['image_dir'], mask_dir=test_dir['mask_dir'], transforms=_EVAL_AUG,
                        CV=dict(k=10, i=-1), training=False, batch_size=1, num_workers=1)
