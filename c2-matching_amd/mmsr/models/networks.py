"""``define_net_*`` factories with the reference's contract (networks.py:4-76): the YAML block's ``type`` names a class
in any ``*_arch.py`` module, the remaining keys are its constructor kwargs."""
from mmsr.models.archs import _arch_modules


def dynamical_instantiation(modules, cls_type, opt):
    for module in modules:
        cls_ = getattr(module, cls_type, None)
        if cls_ is not None:
            return cls_(**opt)
    raise ValueError(f'{cls_type} is not found.')


def _define(opt, key):
    opt_net = opt[key]
    return dynamical_instantiation(_arch_modules, opt_net.pop('type'), opt_net)


def define_net_g(opt):
    return _define(opt, 'network_g')


def define_net_d(opt):
    return _define(opt, 'network_d')


def define_net_map(opt):
    return _define(opt, 'network_map')


def define_net_extractor(opt):
    return _define(opt, 'network_extractor')


def define_net_student(opt):
    return _define(opt, 'network_student')


def define_net_teacher(opt):
    return _define(opt, 'network_teacher')
