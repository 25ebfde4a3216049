"""MI355X-native SVGP/WSVGP hot path behind the gpzoo.kernels / gpzoo.gp API."""
