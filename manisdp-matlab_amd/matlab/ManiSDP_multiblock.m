function [X, obj, data] = ManiSDP_multiblock(At, b, c, K, options)
%MANISDP_MULTIBLOCK  GPU drop-in for the reference's src/primal/ManiSDP_multiblock.m:
%   Min <C, X>  s.t.  A(X) = b,  X = diag(X_1, ..., X_t) >= 0,  diag(X_i) = 1 for i <= K.nob   (K.s = block orders).
%   Same call, option names, defaults, printed lines and data fields (X, S as cell arrays).
%
%   The product manifold of multiblockmanifold.m lives on the GPU as ONE factor: a pmax x sum(K.s) matrix whose
%   column range i holds Y{i} in its first p(i) rows and zeros below (zero rows stay zero under every operation of
%   the trust-region solve).  trustregions(), c'*x, A*x, eS, z and S come from libmanisdp_hip through manisdp_mex;
%   what stays here is the per-block bookkeeping of the outer loop on the small blocks: eig(S{i}), the rank of Y{i},
%   the escape directions.
if nargin < 5, options = struct(); end
nset = reshape(K.s, 1, []);  nb = numel(nset);  nob = K.nob;
defaults = {'min_facsize', 2; 'p0', ones(nb, 1); 'AL_maxiter', 1000; 'gama', 2; 'sigma0', 1e-1; 'sigma_min', 1e-2; ...
            'sigma_max', 1e7; 'tol', 1e-8; 'theta', 1e-2; 'delta', 8; 'alpha', 0.1; 'tolgradnorm', 1e-8; ...
            'TR_maxinner', 20; 'TR_maxiter', 4; 'tau1', 1e1; 'tau2', 1e1; 'line_search', 0};
opt = options;
for q = 1:size(defaults, 1)
    if ~isfield(opt, defaults{q, 1}), opt.(defaults{q, 1}) = defaults{q, 2}; end
end
fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', max(nset), size(b, 1));

first = cumsum([1, nset(1:end-1)]);  last = cumsum(nset);        % column range of block i in the joint factor
cols = arrayfun(@(i) first(i):last(i), 1:nb, 'UniformOutput', false);
p = nset;  big = nset >= opt.min_facsize;  p(big) = opt.p0(big);
h = manisdp_mex('create_multiblock', At, b, c, nset, nob);
release = onCleanup(@() manisdp_mex('destroy', h)); %#ok<NASGU>
unitcols = @(Z) Z./sqrt(sum(Z.^2, 1));
Y = cell(nb, 1);
for i = 1:nb
    Y{i} = randn(p(i), nset(i));
    if i <= nob, Y{i} = unitcols(Y{i}); end
end
U = {};
bvec = full(b(:));  y = zeros(numel(bvec), 1);  sigma = opt.sigma0;  scale_b = 1 + norm(bvec);
data.status = 0;  watch = [];  eta = inf;  t0 = tic;

for iter = 1:opt.AL_maxiter
    tr = struct('maxiter', opt.TR_maxiter, 'maxinner', opt.TR_maxinner, 'tolgradnorm', opt.tolgradnorm, ...
                'Delta_bar', sqrt(pi*sum(nset(1:nob)) + sum(p(nob+1:end).*nset(nob+1:end))));
    manisdp_mex('set_multipliers', h, y, sigma);
    manisdp_mex('set_point', h, side_by_side(Y, p, cols));
    if ~isempty(U)
        Ujoint = side_by_side(U, p, cols);
        base = manisdp_mex('linesearch_cost', h, [], 0);
        step = 1;  tries = 1;
        while tries <= 16 && manisdp_mex('linesearch_cost', h, Ujoint, step) - base > -1e-3
            step = 0.8*step;  tries = tries + 1;
        end
        manisdp_mex('linesearch_accept', h);
    end
    info = manisdp_mex('rtr', h, tr);
    gradnorm = info.gradnorm;
    W = manisdp_mex('get_point', h);
    for i = 1:nb, Y{i} = W(1:p(i), cols{i}); end
    Yeval = Y;

    [obj, Ax] = manisdp_mex('al_primal', h);
    resid = Ax - bvec;
    pinf = norm(resid)/scale_b;
    y = y - sigma*resid;
    z = manisdp_mex('al_dual', h, y);                       % zero on the blocks without a diagonal constraint
    by = bvec'*y + sum(z);
    S = cell(nb, 1);  lowvec = cell(nb, 1);  nneg = zeros(1, nb);  dinfs = zeros(1, nb);
    for i = 1:nb
        S{i} = manisdp_mex('get_dual_slack_block', h, cols{i}(1), numel(cols{i}));   % only the diagonal blocks leave the device
        [V, w] = eig((S{i} + S{i}')/2, 'vector');
        dinfs(i) = max(0, -w(1))/(1 + abs(w(end)));
        nneg(i) = sum(w < 0);  lowvec{i} = V;
    end
    dinf = max(dinfs);
    gap = abs(obj - by)/(abs(by) + abs(obj) + 1);
    fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, p_max:%d, sigma:%0.3f, time:%0.2fs\n', ...
            iter, obj, gap, pinf, dinf, gradnorm, max(p), sigma, toc(t0));
    eta = max([gap, pinf, dinf]);
    if eta < opt.tol, fprintf('Optimality is reached!\n'); break; end
    if mod(iter, 50) == 0
        if iter > 100 && ~isempty(watch) && all([gap, pinf, dinf] > watch)
            data.status = 2;  fprintf('Slow progress!\n');  break;
        end
        watch = [gap, pinf, dinf];
    end

    U = cell(nb, 1);
    for i = find(big)
        if p(i) > 1                                         % numerical rank from the p x p Gram matrix
            G = Y{i}*Y{i}';
            [Q, g] = eig((G + G')/2, 'vector');
            [g, order] = sort(max(g, 0), 'descend');  Q = Q(:, order);
            r = max(sum(sqrt(g) >= opt.theta*sqrt(g(1))), 1);
            if r < p(i), Y{i} = Q(:, 1:r)'*Y{i};  p(i) = r; end
        end
        k = min(nneg(i), opt.delta);
        if i <= nob, k = max(k, 1); end
        if p(i) + k > nset(i), k = 0; end
        D = lowvec{i}(:, 1:k)';
        if opt.line_search == 1
            U{i} = [zeros(p(i), nset(i)); D];  Y{i} = [Y{i}; zeros(k, nset(i))];
        else
            Y{i} = [Y{i}; opt.alpha*D];
            if i <= nob, Y{i} = unitcols(Y{i}); end
        end
        p(i) = p(i) + k;
    end
    if opt.line_search == 1
        for i = find(~big), U{i} = zeros(size(Y{i})); end
    else
        U = {};
    end
    if pinf < opt.tau1*gradnorm
        sigma = max(sigma/opt.gama, opt.sigma_min);
    elseif pinf > opt.tau2*gradnorm
        sigma = min(sigma*opt.gama, opt.sigma_max);
    end
end

X = cellfun(@(Z) Z'*Z, Yeval, 'UniformOutput', false);
data.X = X;  data.y = y;  data.S = S;  data.gap = gap;  data.pinf = pinf;  data.dinf = dinf;
data.gradnorm = gradnorm;  data.time = toc(t0);
if data.status == 0 && eta > opt.tol
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(t0));
end

function W = side_by_side(blocks, p, cols)
% cell array of p(i) x n(i) matrices -> one max(p) x sum(n) matrix, zero rows below each block's own width
W = zeros(max(p), cols{end}(end));
for i = 1:numel(blocks), W(1:size(blocks{i}, 1), cols{i}) = blocks{i}; end
end
