"""The attribute surface of the reference's `st.regda.*` configuration modules (configs/st/regda/2potsdam.py:6-48 and
2vaihingen.py:6-48), built from one table.  `install(module_globals, target)` fills a config module with exactly the
names the reference's entry points read (`cfg.MODEL`, `cfg.SNAPSHOT_DIR`, `cfg.TARGET_DATA_CONFIG`, ...); the
dataset side (directories, normalisation constants, the source / eval / pseudo / test loader settings) comes from
configs.ToPotsdam / configs.ToVaihingen like in the reference."""
import importlib

# optimisation schedule and pseudo-label thresholds: (name, value), identical for both adaptation directions
_SCHEDULE = (
    ('MODEL', 'ResNet101'), ('IGNORE_LABEL', -1), ('MOMENTUM', 0.9), ('WEIGHT_DECAY', 0.0005), ('LEARNING_RATE', 1e-2),
    ('STAGE1_STEPS', 4000), ('STAGE2_STEPS', 6000), ('STAGE3_STEPS', 6000),
    ('NUM_STEPS', None),        # filled by the training script: length of the poly schedule
    ('PREHEAT_STEPS', None),    # filled by the training script: warm-up length
    ('POWER', 0.9), ('EVAL_EVERY', 500), ('GENE_EVERY', 1000), ('CUTOFF_TOP', 0.8), ('CUTOFF_LOW', 0.6),
)
_FROM_DATASET = ('SOURCE_DATA_CONFIG', 'EVAL_DATA_CONFIG', 'PSEUDO_DATA_CONFIG', 'TEST_DATA_CONFIG', 'TARGET_SET',
                 'target_dir', 'DATASETS', 'MEAN', 'STD')


def _target_loader(ds):
    """Unlabelled target crops with stored soft labels (`label_type='prob'`) and SAM region maps (`read_sup`)."""
    augment = [('RandomCrop', (512, 512))]
    augment += [(name, 0.5) for name in ('RandomHorizontalFlip', 'RandomVerticalFlip', 'RandomRotate90')]
    augment.append(('Normalize', dict(mean=ds.MEAN, std=ds.STD, clamp=True)))
    return dict(image_dir=ds.target_dir['image_dir'], mask_dir=[None], transforms=augment, CV=dict(k=10, i=-1),
                training=True, batch_size=8, num_workers=4, pin_memory=True, label_type='prob', read_sup=True)


def install(ns, target):
    """target: 'potsdam' | 'vaihingen'."""
    ds = importlib.import_module('configs.To' + target.capitalize())
    for name in _FROM_DATASET:
        ns[name] = getattr(ds, name)
    ns.update(_SCHEDULE)
    ns['SNAPSHOT_DIR'] = './log/regda/2' + target
    ns['TARGET_DATA_CONFIG'] = _target_loader(ds)
