// The two globals fxcm reads (extern in src/models/fxcmv1.cpp:47), defined in the reference by predictor.cpp:359.
int lstmpr = 0, lstmex = 0;
