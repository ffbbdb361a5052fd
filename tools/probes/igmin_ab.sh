# igemm.hip: the block-count threshold of the channel-tile choice (SV_IG_MIN_TILES) on the decoder layers and the whole step
bash tools/ab.sh igemm.hip "dec[0-5]|conv1x1|s2" "-DSV_IG_MIN_TILES=512" "-DSV_IG_MIN_TILES=256" "-DSV_IG_MIN_TILES=128"
