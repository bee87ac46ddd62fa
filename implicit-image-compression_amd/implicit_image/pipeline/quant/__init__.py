"""K-means weight quantisation ("next" row §8f-1; reference: implicit_image/pipeline/quant/)."""
from .kmeans import KmeansQuant, find_centroids, find_centroids_native, kmeans_fit, kmeans_predict, scatter_mean
from .context import Quantize

__all__ = ["KmeansQuant", "Quantize", "find_centroids", "find_centroids_native", "kmeans_fit", "kmeans_predict", "scatter_mean"]
