"""Flag system of the reference, same names / defaults / quirks (config/base_config.py:12-109).

Kept drop-in: ``BaseConfig().parse()`` returns the argparse Namespace every class of the reference
receives as ``config``. Quirks preserved on purpose: ``type=bool`` flags treat any
non-empty string as True; ``--rendering_zoom_height`` is declared int with a float default.
"""
import argparse
import os

_FLAGS = [
    # name, type, default, help  (reference line)
    ("sim_env", str, "/xmls/acorn_env.xml", "path to simulation environment xml file"),            # :13
    ("verbose", bool, False, "whether to show config information"),                                  # :15
    ("width_capture", int, 64, "width of the image to be captured"),                                # :18
    ("height_capture", int, 64, "height of the image to be captured"),                              # :19
    ("rendering_zoom_width", int, 5 * 2, "width of the rendering zoom"),                            # :20
    ("rendering_zoom_height", int, 3.75 * 2, "height of the rendering zoom"),                       # :21
    ("full_observation", bool, True, "True for RGBD observation, False for RGB observation"),       # :22
    ("camera_id", int, 3, "workbench_camera: 0, upper_camera: 1, gripper_camera: 2, all: 3"),       # :24
    ("show_obs", bool, False, "True for live rendering observations"),                              # :26
    ("max_rotation", float, 0.15, "maximum rotation of the gripper"),                               # :29
    ("max_translation", float, 0.05, "maximum translation of the gripper"),                         # :30
    ("grasp_tolerance", float, 0.03, "joint grasp tolerance"),                                      # :31
    ("pos_tolerance", float, 0.002, "joint position tolerance"),                                    # :32
    ("include_roll", bool, True, "True for including roll in the action"),                          # :33
    ("max_steps", int, 400, "maximum number of timesteps to execute an action"),                    # :36
    ("im_reward", bool, False, "True for adding intrinsic reward"),                                 # :38
    ("her_buffer", bool, False, "True for adding HER buffer"),                                      # :39
    ("direction", int, 0, "target vector direction"),                                               # :42
    ("time_horizon", int, 400, "maximum number of steps per episode"),                              # :45
    ("trained_models", str, "/models/trained_models", "path to trained models"),                    # :48
    ("name", str, "SAC", "name of the experiment"),                                                 # :50
    ("suffix", str, "", "customized suffix: config.name = config.name + suffix"),                   # :51
]


class BaseConfig:
    def __init__(self):
        self.initialized = False
        self.parser = None

    def initialize(self, parser):
        for name, typ, default, hlp in _FLAGS:
            parser.add_argument(f"--{name}", type=typ, default=default, help=hlp)
        self.initialized = True
        return parser

    def gather_config(self, argv=None):
        if not self.initialized:
            parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
            self.parser = self.initialize(parser)
        return self.parser.parse_args(argv)

    @staticmethod
    def _apply_suffix(config):
        if config.suffix:
            config.name = config.name + "_" + config.suffix.format(**vars(config))

    def print_config(self, config):
        lines = ["---------------- Config -----------------"]
        for k, v in sorted(vars(config).items()):
            default = self.parser.get_default(k)
            comment = "\t[default: %s]" % str(default) if v != default else ""
            lines.append("{:>25}: {:<30}{}".format(str(k), str(v), comment))
        lines.append("----------------- End -------------------")
        message = "\n".join(lines)
        print(message)
        self._apply_suffix(config)      # the reference applies the suffix a second time here (:80-83)
        expr_dir = os.path.join(config.trained_models, config.name)
        os.makedirs(expr_dir, exist_ok=True)
        with open(os.path.join(expr_dir, "config.txt"), "wt") as f:
            f.write(message + "\n")

    def parse(self, argv=None):
        config = self.gather_config(argv)
        self._apply_suffix(config)
        self.config = config
        return config
