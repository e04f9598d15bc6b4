P=./tools/native/dma_probe
echo "strided 64-B pieces (BK=32 pattern), 310 MB matrix, 12 k-tiles:"
$P 64 40960 1576 12 403456 1
$P 64 40960 1576 12 403456 2
echo "strided 128-B pieces (BK=64 pattern), 6 k-tiles:"
$P 128 81920 1576 6 403456 1
echo "contiguous 16 KB chunks (one k-tile, ld = row bytes):"
$P 128 40960 9456 1 2420736 1
$P 64 40960 18912 1 4841472 1
