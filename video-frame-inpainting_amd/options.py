"""argparse flag system of the path (reference src/options/options.py:6-209): same flag names, defaults and
post-processing (image_size / padding_size widened to two-element lists, :54-58; ``assert torch.cuda.is_available()``,
:61).  Flags that only configure the reference's other model families (transformer teacher forcing, SloMo weights) and
the video-list paths of the real datasets are accepted where the reference requires them but are optional here: the
configs of this build run on synthetic clips (``--synthetic`` is the one added flag)."""
import argparse

import torch


class BaseOptions(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        g = self.parser.add_argument_group('Experiment parameters')
        g.add_argument('--name', type=str, default='experiment_name', help='Name of the experiment')
        g = self.parser.add_argument_group('Model input/output parameters')
        g.add_argument('--K', type=int, required=True, help='Length of the preceding sequence (in frames)')
        g.add_argument('--T', type=int, required=True, help='Length of the middle sequence (in frames)')
        g.add_argument('--F', type=int, required=True, help='Length of the following sequence (in frames)')
        g.add_argument('--batch_size', type=int, default=4, help='Mini-batch size')
        g.add_argument('--image_size', type=int, nargs='+', default=[128], help='Image size (H x W); one number means H = W')
        g.add_argument('--padding_size', type=int, nargs='+', default=[0],
                       help='Padding added to the bottom and right sides of the image; one number means both')
        g.add_argument('--c_dim', type=int, default=3, help='Number of channels in the image input')
        g = self.parser.add_argument_group('Model specification parameters')
        g.add_argument('--model_key', type=str, required=True, help='Key identifying the generator to create')
        g = self.parser.add_argument_group('Directory parameters')
        g.add_argument('--checkpoints_dir', type=str, default='checkpoints', help='Path to store/load checkpoint files')
        g = self.parser.add_argument_group('Common data loading parameters')
        g.add_argument('--num_threads', type=int, default=2, help='Number of threads used to load data')
        g.add_argument('--synthetic', type=int, default=0, metavar='N_CLIPS',
                       help='(this build) run on N seeded synthetic clips instead of a video list')
        g.add_argument('--seed', type=int, default=1002, help='(this build) seed of the synthetic clips')
        g.add_argument('--winograd_arithmetic', type=str, default='fp32', choices=['fp32', 'bf16x3'],
                       help='(this build) arithmetic of the 3x3 Winograd GEMMs: fp32 = the fp32 MFMA (default, the arithmetic every '
                            'parity statement is made on); bf16x3 = opt-in split bf16 (three bf16 terms per operand, six products, '
                            'fp32 accumulation: error at or below the fp32 form, 3-25 %% faster per layer)')

        g.add_argument('--winograd_tile', type=int, default=4, choices=[2, 4],
                       help='(this build) output tile of the 3x3 Winograd convolutions at inference: 4 = F(4x4,3x3) on the layers with at least '
                            "128 input and output channels (default: 1.78x fewer MFMAs there, the forward's end-to-end error unchanged, ~3 %% "
                            'faster at 32 clips) and F(2x2,3x3) elsewhere; 2 = F(2x2,3x3) on every layer')

    def parse(self, args=None, allow_unknown=False, require_gpu=True):
        if allow_unknown:
            opt, unknown_opt = self.parser.parse_known_args(args)
            print('Ignored arguments: %s' % str(unknown_opt))
        else:
            opt = self.parser.parse_args(args)
        if len(opt.image_size) == 1:
            opt.image_size.append(opt.image_size[0])
        if len(opt.padding_size) == 1:
            opt.padding_size.append(opt.padding_size[0])
        if require_gpu:
            assert torch.cuda.is_available()      # options.py:61: the path has no CPU implementation
        return opt


class TrainOptions(BaseOptions):
    def __init__(self):
        super().__init__()
        g = self.parser.add_argument_group('Optimization parameters')
        g.add_argument('--lr', type=float, default=0.0001, help='Base learning rate')
        g.add_argument('--beta1', type=float, default=0.5, help='Momentum term of adam')
        g.add_argument('--max_iter', type=int, default=100000, help='Maximum number of iterations (batches) to train on')
        g = self.parser.add_argument_group('Loss parameters')
        g.add_argument('--alpha', type=float, default=1.0, help='Image loss weight')
        g.add_argument('--beta', type=float, default=0.02, help='GAN loss weight')
        g = self.parser.add_argument_group('Training frequency parameters')
        g.add_argument('--print_freq', type=int, default=100)
        g.add_argument('--save_latest_freq', type=int, default=1000)
        g.add_argument('--validate_freq', type=int, default=10000)
        g = self.parser.add_argument_group('Adversarial training parameters')
        g.add_argument('--df_dim', type=int, default=64, help='Number of filters in first conv layer of the discriminator')
        g.add_argument('--Ip', type=int, default=3, help='Power iterations of the spectral-normalized discriminator')
        g.add_argument('--disc_window_size', type=int, default=3, help='Frames the discriminator sees at a time')
        g = self.parser.add_argument_group('Training data loading parameters')
        g.add_argument('--alt_K', type=int, default=None)
        g.add_argument('--alt_T', type=int, default=None)
        g.add_argument('--alt_F', type=int, default=None)
        for name in ('train_video_list_path', 'val_video_list_path', 'val_video_list_alt_T_path',
                     'val_video_list_alt_K_F_path', 'vis_video_list_path', 'vis_video_list_alt_T_path',
                     'vis_video_list_alt_K_F_path'):
            g.add_argument('--' + name, type=str, default=None)
        g.add_argument('--serial_batches', action='store_true')
        g.add_argument('--no_backwards', action='store_true')
        g.add_argument('--no_flip', action='store_true')
        g.add_argument('--sample_KTF', action='store_true',
                       help='Sample the number of preceding, middle, and following frames in each minibatch')
        g.add_argument('--graph_step', action='store_true',
                       help='Capture one whole update as a hipGraph per (K, T, F) and replay it (one process per node only)')
        g.add_argument('--miopen_find_mode', type=str, default=None, choices=['NORMAL', 'FAST', 'HYBRID', 'DYNAMIC_HYBRID'],
                       help='MIOPEN_FIND_MODE for this run (package default FAST: first update 1.4 s instead of 22 s, later updates '
                            '2-3 %% slower; NORMAL pays the search once and is the choice for a long run)')
        g = self.parser.add_argument_group('Training visualization parameters')
        g.add_argument('--tensorboard_dir', type=str, default='tb')


class TestOptions(BaseOptions):
    def __init__(self):
        super().__init__()
        g = self.parser.add_argument_group('Test data loading parameters')
        g.add_argument('--test_video_list_path', type=str, default=None,
                       help='The path to the text file containing the list of test video (clips)')
        g.add_argument('--disjoint_clips', action='store_true')
        g = self.parser.add_argument_group('Snapshot parameters')
        g.add_argument('--snapshot_file_name', type=str, default='model_best.ckpt')
        g = self.parser.add_argument_group('Qualitative result destination parameters')
        g.add_argument('--qual_result_root', type=str, required=True,
                       help='The root path where qualitative results will be stored')
        g = self.parser.add_argument_group('Output parameters')
        g.add_argument('--intermediate_preds', action='store_true',
                       help='Flag to write intermediate predictions in addition to final ones')
        g.add_argument('--random_init', action='store_true',
                       help='(this build) skip the snapshot load and keep the seeded xavier initialisation')
