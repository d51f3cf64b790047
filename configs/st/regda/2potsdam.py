"""st.regda.2potsdam -- the attribute surface of the reference's configs/st/regda/2potsdam.py:6-48."""
from configs.ToPotsdam import (SOURCE_DATA_CONFIG, EVAL_DATA_CONFIG, PSEUDO_DATA_CONFIG, TEST_DATA_CONFIG,  # noqa: F401
                               TARGET_SET, target_dir, DATASETS, MEAN, STD)

MODEL = 'ResNet101'

IGNORE_LABEL = -1
MOMENTUM = 0.9

SNAPSHOT_DIR = './log/regda/2potsdam'

# Hyper Paramters
WEIGHT_DECAY = 0.0005
LEARNING_RATE = 1e-2
STAGE1_STEPS = 4000
STAGE2_STEPS = 6000
STAGE3_STEPS = 6000
NUM_STEPS = None        # for learning rate poly
PREHEAT_STEPS = None    # for warm-up
POWER = 0.9             # lr poly power
EVAL_EVERY = 500
GENE_EVERY = 1000
CUTOFF_TOP = 0.8
CUTOFF_LOW = 0.6

TARGET_DATA_CONFIG = dict(
    image_dir=target_dir['image_dir'],
    mask_dir=[None],
    transforms=[('RandomCrop', (512, 512)), ('RandomHorizontalFlip', 0.5), ('RandomVerticalFlip', 0.5),
                ('RandomRotate90', 0.5), ('Normalize', dict(mean=MEAN, std=STD, clamp=True))],
    CV=dict(k=10, i=-1),
    training=True,
    batch_size=8,
    num_workers=4,
    pin_memory=True,
    label_type='prob',
    read_sup=True,
)
