"""Stand-in for ConfigArgParse as the reference's config_parser.py:1-66 uses it: an argparse parser with one
``is_config_file`` option naming a YAML file of ``key: value`` lines whose values become defaults that the command line
overrides.  TEST INFRASTRUCTURE ONLY (tests/test_reference_entry_script.py)."""
import argparse

import yaml

ArgumentDefaultsHelpFormatter = argparse.ArgumentDefaultsHelpFormatter


class DefaultConfigFileParser(object):
    pass


class ArgParser(argparse.ArgumentParser):
    def __init__(self, *a, config_file_parser_class=None, **kw):
        super(ArgParser, self).__init__(*a, **kw)
        self._config_dest = None

    def add_argument(self, *a, is_config_file=False, **kw):
        act = super(ArgParser, self).add_argument(*a, **kw)
        if is_config_file:
            self._config_dest = act.dest
        return act

    def parse_known_args(self, args=None, namespace=None):
        first, _ = super(ArgParser, self).parse_known_args(args, namespace)
        path = getattr(first, self._config_dest) if self._config_dest else None
        if path:
            with open(path) as f:
                cfg = yaml.safe_load(f) or {}
            known = {a.dest: a for a in self._actions}
            defaults = {}
            for k, v in cfg.items():
                if v is None or k not in known:
                    continue
                t = known[k].type
                defaults[k] = t(v) if callable(t) else v
            self.set_defaults(**defaults)
        return super(ArgParser, self).parse_known_args(args, namespace)
