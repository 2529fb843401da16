"""Command-line configuration -- same flags, dests and defaults as the reference's config.py:9-63 (parsed at import,
exposed as `cfg`), plus additive flags of this MI355X implementation (precision, synthetic data, launch options)."""
import argparse

parser = argparse.ArgumentParser('Configurations for SPEED+ Baseline Study (MI355X implementation)')

# basic directories and names
parser.add_argument('--seed', type=int, default=2021)
parser.add_argument('--projroot', type=str, default='.')
parser.add_argument('--dataroot', type=str, default='data')
parser.add_argument('--dataname', type=str, default='speedplus')
parser.add_argument('--savedir', type=str, default='checkpoints/synthetic/krn')
parser.add_argument('--resultfn', type=str, default='')
parser.add_argument('--logdir', type=str, default='log/synthetic/krn')
parser.add_argument('--pretrained', type=str, default='')

# model
parser.add_argument('--model_name', type=str, default='krn')
parser.add_argument('--input_shape', nargs='+', type=int, default=(224, 224))
parser.add_argument('--num_keypoints', type=int, default=11)
parser.add_argument('--num_classes', type=int, default=5000)
parser.add_argument('--num_neighbors', type=int, default=5)
parser.add_argument('--keypts_3d_model', type=str, default='src/utils/tangoPoints.mat')
parser.add_argument('--attitude_class', type=str, default='src/utils/attitudeClasses.mat')

# training
parser.add_argument('--start_over', dest='auto_resume', action='store_false', default=True)
parser.add_argument('--randomize_texture', dest='randomize_texture', action='store_true', default=False)
parser.add_argument('--perform_dann', dest='dann', action='store_true', default=False)
parser.add_argument('--texture_alpha', type=float, default=0.5)
parser.add_argument('--texture_ratio', type=float, default=0.5)
parser.add_argument('--use_fp16', dest='fp16', action='store_true', default=False,
                    help='reference: fp16 autocast + GradScaler; here: KRN and SPN run the IEEE-half build of the kernels with GradScaler\'s dynamic loss scaling kept on the device; RevGrad / DANN (adapt.py has no mixed precision in the reference) runs bfloat16 (logged); --precision bf16 selects bfloat16 for KRN / SPN')
parser.add_argument('--batch_size', type=int, default=32)
parser.add_argument('--max_epochs', type=int, default=75)
parser.add_argument('--num_workers', type=int, default=8)
parser.add_argument('--test_epoch', type=int, default=-1)
parser.add_argument('--optimizer', type=str, default='rmsprop')
parser.add_argument('--lr', type=float, default=0.001)
parser.add_argument('--momentum', type=float, default=0.9)
parser.add_argument('--weight_decay', type=float, default=5e-5)
parser.add_argument('--lr_decay_alpha', type=float, default=0.96)
parser.add_argument('--lr_decay_step', type=int, default=1)

# datasets
parser.add_argument('--train_domain', type=str, default='synthetic')
parser.add_argument('--test_domain', type=str, default='lightbox')
parser.add_argument('--train_csv', type=str, default='train.csv')
parser.add_argument('--test_csv', type=str, default='lightbox.csv')

# misc
parser.add_argument('--gpu_id', type=int, default=0)
parser.add_argument('--no_cuda', dest='use_cuda', action='store_false', default=True)

# ---- additions of this implementation (all optional)
parser.add_argument('--precision', type=str, default=None, choices=[None, 'fp32', 'bf16', 'fp16'],
                    help='compute type of the HIP kernels; default fp32 (the reference trains in fp32); --use_fp16 selects fp16 (IEEE half + device-side '
                         'GradScaler) for SPN and KRN, bf16 for RevGrad / DANN')
parser.add_argument('--styleaug_precision', type=str, default='bf16', choices=['bf16', 'fp16', 'fp32'],
                    help='style decoder (--randomize_texture): bf16 matrix-core kernels (default), the same kernels in IEEE half (+2 % time, '
                         '8x closer to the float32 image), or the float32 reference-precision mode (the reference runs this module in fp32, '
                         'trainer.py:68-69; ~30x slower)')
parser.add_argument('--deterministic', dest='deterministic', action='store_true', default=False,
                    help='KRN / RevGrad on the reproducible build of the kernels (exact, order-independent accumulation instead of float '
                         'atomics): the same seed gives bit-identical checkpoints in every run.  ~2x slower; the reference has no such mode '
                         '(utils.py:297-298: cudnn.deterministic = False)')
parser.add_argument('--synthetic_batches', type=int, default=0,
                    help='>0: train/evaluate on this many synthetic batches per epoch (U[0,1) images, U[0,1) keypoints); '
                         'the SPEED+ dataset pipeline is not part of this build')

cfg = parser.parse_args()
