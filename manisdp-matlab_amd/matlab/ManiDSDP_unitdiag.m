function [X, obj, data] = ManiDSDP_unitdiag(A, b, c, K, options)
%MANIDSDP_UNITDIAG  GPU drop-in for the reference's src/dual/ManiDSDP_unitdiag.m (dual approach):
%   sup <C, X> + <c_f, w>  s.t.  A(X) + B(w) = b,  X >= 0,  w free, where the dual slack S has diag(S) = 1.
%   A is m x (K.f + K.s^2) with the K.f free columns first, c likewise.  Same call, option names, defaults
%   (options.dAAt = diag(A_psd*A_psd') is computed when absent), printed lines and data fields.
%
%   The Riemannian subproblem over the oblique factor of S -- cost, gradient, Hess-vec, trustregions(), the line
%   search -- and the outer-step algebra (y, As, the update of the n^2 multiplier x, eX, z, X) run inside
%   libmanisdp_hip; x never leaves the GPU.  This file keeps the scalar bookkeeping of the outer loop, eig(X) and
%   the rank decision on the p x p Gram matrix of the factor.
if nargin < 5, options = struct(); end
m = size(b, 1);  n = K.s;  nf = 0;
if isfield(K, 'f'), nf = K.f; end
defaults = {'p0', ceil(log(m)); 'ADMM_maxiter', 300; 'gama', 2; 'sigma0', 1e-3; 'sigma_min', 1e-3; 'sigma_max', 1e7; ...
            'tol', 1e-8; 'theta', 1e-3; 'delta', 8; 'alpha', 0.1; 'tolgradnorm', 1e-8; 'TR_maxinner', 20; ...
            'TR_maxiter', 4; 'tau1', 1e1; 'tau2', 1e2; 'line_search', 0};
opt = options;
for q = 1:size(defaults, 1)
    if ~isfield(opt, defaults{q, 1}), opt.(defaults{q, 1}) = defaults{q, 2}; end
end
fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', n, m);

cost_scale = 1 + norm(c);
Bfree = A(:, 1:nf);  Apsd = A(:, nf+1:end);
cfree = full(c(1:nf));  cpsd = c(nf+1:end);
if ~isfield(opt, 'dAAt'), opt.dAAt = full(sum(Apsd.^2, 2)); end
h = manisdp_mex('create_dual_unitdiag', Apsd', opt.dAAt, b, cpsd, n, sparse(Bfree), cfree);
release = onCleanup(@() manisdp_mex('destroy', h)); %#ok<NASGU>

p = opt.p0;  sigma = opt.sigma0;  w = zeros(nf, 1);
Y = randn(p, n);  Y = Y./sqrt(sum(Y.^2, 1));
U = [];
tr = struct('maxiter', opt.TR_maxiter, 'maxinner', opt.TR_maxinner, 'tolgradnorm', opt.tolgradnorm);
data.status = 0;  widths = [];  history = [];  watch = [];  eta = inf;  t0 = tic;

for iter = 1:opt.ADMM_maxiter
    widths(end+1, 1) = p; %#ok<AGROW>
    manisdp_mex('dual_set_penalty', h, sigma, w);
    manisdp_mex('set_point', h, Y);
    if ~isempty(U)
        base = manisdp_mex('linesearch_cost', h, [], 0);
        step = 1;  tries = 1;
        while tries <= 16 && manisdp_mex('linesearch_cost', h, U, step) - base > -1e-3
            step = 0.8*step;  tries = tries + 1;
        end
        manisdp_mex('linesearch_accept', h);
    end
    info = manisdp_mex('rtr', h, tr);
    gradnorm = info.gradnorm;
    Y = manisdp_mex('get_point', h);
    Yeval = Y;

    [by, cex, as2, Af, z] = manisdp_mex('dual_outer_step', h);     % x <- x - sigma*As happens on the device
    pinf = (sqrt(as2) + norm(Af))/cost_scale;
    w = w - sigma*Af;
    obj = cex + cfree'*w + sum(z);
    X = manisdp_mex('get_dual_slack', h);                           % eX - diag(z)
    [vecs, vals] = eig((X + X')/2, 'vector');
    dinf = max(0, -vals(1))/(1 + abs(vals(end)));
    gap = abs(obj - by)/(1 + abs(obj) + abs(by));

    [Q, sv2] = eig(Y*Y', 'vector');                                 % singular values of Y from its p x p Gram matrix
    [sv2, order] = sort(max(sv2, 0), 'descend');  Q = Q(:, order);
    sv = sqrt(sv2);
    r = sum(sv > opt.theta*sv(1));
    fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs\n', ...
            iter, obj, gap, pinf, dinf, gradnorm, r, p, sigma, toc(t0));
    eta = max([gap, pinf, dinf]);
    history(end+1, 1) = eta; %#ok<AGROW>
    if eta < opt.tol
        fprintf('Optimality is reached!\n');
        break;
    end
    if mod(iter, 50) == 0
        if iter > 100 && all([gap, pinf, dinf] > watch)
            data.status = 2;
            fprintf('Slow progress!\n');
            break;
        end
        watch = [gap, pinf, dinf];
    end
    if r <= p - 1
        Y = Q(:, 1:r)'*Y;                                           % = V(:,1:r)'.*e(1:r) of svd(Y)
        p = r;
    end
    nne = max(min(sum(vals < 0), opt.delta), 1);
    if opt.line_search == 1
        U = [zeros(p, n); vecs(:, 1:nne)'];
        Y = [Y; zeros(nne, n)]; %#ok<AGROW>
    else
        Y = [Y; opt.alpha*vecs(:, 1:nne)']; %#ok<AGROW>
        Y = Y./sqrt(sum(Y.^2, 1));
    end
    p = p + nne;
    if pinf < opt.tau1*gradnorm
        sigma = max(sigma/opt.gama, opt.sigma_min);
    elseif pinf > opt.tau2*gradnorm
        sigma = min(sigma*opt.gama, opt.sigma_max);
    end
end

data.X = X;
data.y = manisdp_mex('dual_get_y', h);
data.S = Yeval'*Yeval;
data.w = w;
data.gap = gap;  data.pinf = pinf;  data.dinf = dinf;  data.gradnorm = gradnorm;
data.time = toc(t0);
data.fac_size = widths;
data.seta = history;
if data.status == 0 && eta > opt.tol
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiDSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(t0));
end
