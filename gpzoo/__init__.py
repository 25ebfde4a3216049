"""Drop-in import name: ``from gpzoo.gp import WSVGP`` etc. resolve to gpzoo_amd."""
